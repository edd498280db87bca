#!/bin/bash
# One GPU-box visit that regenerates everything under profiles/rNN: kernel stats (rocprofv3 --kernel-trace --stats)
# of the three bench workloads, per-layer tables, HBM traffic (separate --pmc passes).  Copy step runs on the host.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
bash tools/prof_stats.sh resnet50 vit_b16 swin_b > gpurun_out/prof.log 2>&1
for wl in resnet50 vit_b16; do timeout -k 10 300 python3 tools/layer_times.py $wl 256 > gpurun_out/layers_$wl.txt 2>&1; done
timeout -k 10 300 python3 tools/layer_times.py swin_b 128 > gpurun_out/layers_swin_b.txt 2>&1
for wl in resnet50 vit_b16 swin_b; do bash tools/pmc_traffic.sh $wl > gpurun_out/traffic_$wl.log 2>&1; tail -1 gpurun_out/traffic_$wl.log | cut -c1-300; done
