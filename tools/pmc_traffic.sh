#!/bin/bash
# HBM traffic of the implicit-GEMM kernel family for one bench.py workload: two separate --pmc passes
# (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md "HBM": FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
WL=${1:-resnet50}
OUT=gpurun_out/traffic_$WL
rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python3 bench.py --workload $WL --steps 3 --warmup 2 --no-cpu-baseline > $OUT/$c.log 2>&1
  echo "pass $c rc=$?"
done
python3 - "$WL" <<'PY'
import csv, glob, json, sys, collections
wl = sys.argv[1]
tot = {}
n = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    s = 0.0; k = 0
    for f in glob.glob(f"gpurun_out/traffic_{wl}/{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if any(k in row["Kernel_Name"] for k in ("conv_igemm", "gemm256", "gemm_pp", "gemm_stream")) and row["Counter_Name"] == c:
                s += float(row["Counter_Value"]); k += 1
    tot[c] = s; n[c] = k
launches = n["FETCH_SIZE"]
fetch_kb, write_kb = tot["FETCH_SIZE"], tot["WRITE_SIZE"]
res = {"workload": wl, "kernel": "conv_igemm_kernel + gemm256 / gemm_pp / gemm_stream kernels (all instantiations)", "launches": launches,
       "FETCH_SIZE_KB_raw_per_launch": fetch_kb / max(launches, 1), "WRITE_SIZE_KB_per_launch": write_kb / max(n["WRITE_SIZE"], 1),
       "gfx950_fetch_correction": 2.0,
       "hbm_bytes_per_launch": (2.0 * fetch_kb / max(launches, 1) + write_kb / max(n["WRITE_SIZE"], 1)) * 1024}
print(json.dumps(res))
open(f"gpurun_out/traffic_{wl}/traffic.json", "w").write(json.dumps(res, indent=1))
PY
