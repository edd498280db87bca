#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py (no PMC), summary CSVs copied to gpurun_out/prof_<name>/
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for wl in "$@"; do
  out=gpurun_out/prof_$wl
  rm -rf $out; mkdir -p $out
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-probe > $out/bench.log 2>&1
  echo "prof $wl rc=$?"; tail -n 2 $out/bench.log | cut -c1-400
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -25 "$f" | cut -c1-220
done
