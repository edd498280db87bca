# One GPU-box visit for everything under profiles/rNN: kernel stats, layer tables, traffic, utilisation counters, the default bench line.
# Host side afterwards: tools/collect_profiles.sh rNN (+ cp gpurun_out/util_*/util.txt profiles/rNN/util_*.txt).
bash tools/refresh_profiles.sh
for wl in resnet50 vit_b16 swin_b; do bash tools/pmc_util.sh $wl > gpurun_out/util_$wl.log 2>&1; tail -n 2 gpurun_out/util_$wl.log | cut -c1-200; done
python bench.py > gpurun_out/bench.log 2>&1; tail -n 1 gpurun_out/bench.log | cut -c1-300
