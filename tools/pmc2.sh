#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
SH=${1:-qkv}
OUT=gpurun_out/pmc2
rm -rf $OUT; mkdir -p $OUT
run() { local tag=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$tag -- python3 tools/conv_micro.py $SH 3 > $OUT/$tag.log 2>&1; echo "pass $tag rc=$?"; }
run a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS
run c SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
run d GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run e TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr
python3 - <<'PY'
import csv, glob, collections
for tag in "abcde":
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/pmc2/{tag}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if any(k in row["Kernel_Name"] for k in ("conv_igemm", "attn_mfma", "conv_halo", "gemm_pp", "gemm_stream")):
                agg[row["Grid_Size"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for g, cs in agg.items():
        print(tag, g, {c: round(sum(v)/len(v)) for c, v in cs.items()})
PY
