#!/bin/bash
# rocprofv3 PMC passes (one counter group per run) over tools/conv_micro.py; CSVs under gpurun_out/pmc
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
SH=${1:-expand56,c3x3_14}
OUT=gpurun_out/pmc
mkdir -p $OUT
run() { # tag counters...
  local tag=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$tag -- python3 tools/conv_micro.py $SH 3 > $OUT/$tag.log 2>&1
  echo "pass $tag rc=$?"
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA
run sq2 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run grbm GRBM_GUI_ACTIVE
python3 - <<'PY'
import csv, glob, collections
for tag in ("sq1","sq2","tcc1","tcc2","grbm"):
    files = glob.glob(f"gpurun_out/pmc/{tag}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name","")
            if "conv_igemm" not in k: continue
            key = (k[:60], row.get("Grid_Size"), row.get("LDS_Block_Size"))
            agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for key, cs in agg.items():
        print(tag, key, {c: sum(v)/len(v) for c, v in cs.items()})
PY
