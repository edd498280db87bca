#!/bin/bash
# MFMA utilisation and LDS bank conflicts per kernel for one bench.py workload: separate --pmc passes (derived counters MfmaUtil,
# LdsBankConflict; MI355X_MICROARCH.md: counters in their own run, with --kernel-trace only).  -> gpurun_out/util_<workload>/util.txt
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
WL=${1:-resnet50}
OUT=gpurun_out/util_$WL
rm -rf $OUT; mkdir -p $OUT
for c in MfmaUtil LdsBankConflict; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python3 bench.py --workload $WL --steps 3 --warmup 2 --no-cpu-baseline --no-also --no-probe > $OUT/$c.log 2>&1
  echo "pass $c rc=$?"
done
python3 - "$WL" <<'PY'
import csv, glob, sys, collections
wl = sys.argv[1]
agg = collections.defaultdict(lambda: {"n": 0})
for c in ("MfmaUtil", "LdsBankConflict"):
    for fn in glob.glob(f"gpurun_out/util_{wl}/{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn)):
            if row["Counter_Name"] != c:
                continue
            a = agg[row["Kernel_Name"][:96]]
            a[c] = a.get(c, 0.0) + float(row["Counter_Value"])
            a["n_" + c] = a.get("n_" + c, 0) + 1
import os
sys.path.insert(0, os.getcwd())
import bench
lines = [f"{wl}: per-kernel averages over the launches of one bench.py run (rocprofv3 --pmc, one counter per pass); csrc_sha {bench.csrc_sha()}",
         "NOTE: the forward runs as two half batches on two streams; under rocprofv3 --pmc the streams are SERIALISED and each half-batch launch of the",
         "      conv / linear family is planned for 128 of the 256 CUs (TLXMI_PLAN_SHARED_HALF: ResNet-50, ViT-B/16) — a chip-wide percentage such as MfmaUtil",
         "      is therefore about HALF of what the two concurrent halves reach together (layer tables: profiles/*/…_layer_times.txt).",
         f"{'launches':>8s} {'MfmaUtil %':>11s} {'LdsBankConflict %':>18s}  kernel"]
for k, a in sorted(agg.items(), key=lambda kv: -kv[1].get("n_MfmaUtil", 0) * kv[1].get("MfmaUtil", 0.0)):
    n = a.get("n_MfmaUtil", 0)
    if not n:
        continue
    mu = a.get("MfmaUtil", 0.0) / n
    lb = a.get("LdsBankConflict", 0.0) / max(a.get("n_LdsBankConflict", 1), 1)
    lines.append(f"{n:8d} {mu:11.1f} {lb:18.2f}  {k}")
open(f"gpurun_out/util_{wl}/util.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:24]))
PY
