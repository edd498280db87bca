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
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python3 bench.py --workload $WL --steps 3 --warmup 2 --no-cpu-baseline --no-also --no-probe > $OUT/$c.log 2>&1
  echo "pass $c rc=$?"
done
python3 - "$WL" <<'PY'
import csv, glob, json, sys
sys.path.insert(0, ".")
import bench
wl = sys.argv[1]
tot, n, fwd = {}, {}, {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    s = 0.0; k = 0; f = 0
    for fn in glob.glob(f"gpurun_out/traffic_{wl}/{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn)):
            if row["Counter_Name"] != c:
                continue
            s += float(row["Counter_Value"]); k += 1      # every kernel of the run: all of them are forwards of the model
            if any(m in row["Kernel_Name"] for m in ("to_nhwc_s2d", "patchify_kernel", "patch_embed4_kernel")):
                f += 1          # one layout transform / patch pass per forward and stream: counts the forwards of the run
    tot[c] = s; n[c] = k; fwd[c] = f
# ResNet-50 / ViT-B/16 (batch 256) and Swin-B (batch 128) run as two half batches on two streams (engine.two_streams): two
# transforms a forward
parts = 2
forwards = max(fwd["FETCH_SIZE"] // parts, 1)
fetch_kb, write_kb = tot["FETCH_SIZE"] / forwards, tot["WRITE_SIZE"] / max(fwd["WRITE_SIZE"] // parts, 1)
res = {"workload": wl, "kernel": "every kernel of a forward", "csrc_sha": bench.csrc_sha(),
       "forwards": forwards, "kernel_launches_per_forward": n["FETCH_SIZE"] / forwards,
       "FETCH_SIZE_KB_raw_per_forward": fetch_kb, "WRITE_SIZE_KB_per_forward": write_kb,
       "gfx950_fetch_correction": 2.0,
       "hbm_bytes_per_forward": (2.0 * fetch_kb + write_kb) * 1024}
print(json.dumps(res))
open(f"gpurun_out/traffic_{wl}/traffic.json", "w").write(json.dumps(res, indent=1))
PY
