#!/bin/bash
# Host side, after `gpurun -- bash tools/refresh_profiles.sh` (and `tools/gpu_ci.sh bench`): copy the
# summaries to be judged from the scratch gpurun_out/ into profiles/<round>/.  gpurun MERGES into gpurun_out/, so older
# rocprof run directories may still be there: always take the newest file.
set -u
R=${1:-r01}
cd "$(dirname "$0")/.."
mkdir -p profiles/$R
: > profiles/$R/bench_lines_under_rocprof.jsonl
for wl in resnet50 vit_b16 swin_b; do
  f=$(find gpurun_out/prof_$wl -name "*kernel_stats.csv" -printf "%T@ %p\n" | sort -n | tail -1 | cut -d" " -f2)
  cp "$f" profiles/$R/${wl}_kernel_stats.csv
  cp gpurun_out/layers_$wl.txt profiles/$R/${wl}_layer_times.txt
  cp gpurun_out/traffic_$wl/traffic.json profiles/$R/traffic_$wl.json
  grep '^{"metric"' gpurun_out/prof_$wl/bench.log >> profiles/$R/bench_lines_under_rocprof.jsonl
  [ -f gpurun_out/util_$wl/util.txt ] && cp gpurun_out/util_$wl/util.txt profiles/$R/util_$wl.txt      # tools/pmc_util.sh
done
if [ -f gpurun_out/bench.log ]; then
  : > profiles/$R/bench_lines.jsonl
  grep '^{"metric"' gpurun_out/bench.log >> profiles/$R/bench_lines.jsonl     # the default run: ResNet-50 + also[ViT-B/16, Swin-B]
fi
ls -la profiles/$R
