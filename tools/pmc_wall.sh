#!/bin/bash
# Which pipe bounds the persistent 256 x 256 GEMM ("replace the wall paragraph with counters"): rocprofv3 --pmc passes
# over tools/wall_micro.py (counters in their own runs, --kernel-trace only), per launch kind: MFMA pipe, LDS array, vector-memory
# and VALU activity as a share of the kernel's own duration.  -> gpurun_out/pmc_wall/wall.txt
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_wall
rm -rf $OUT; mkdir -p $OUT
python3 tools/wall_micro.py 6 > $OUT/unprofiled.txt 2>&1
REPS=2
run() { local tag=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$tag -- python3 tools/wall_micro.py $REPS > $OUT/$tag.log 2>&1
  local rc=$?; echo "pass $tag rc=$rc"
  if [ $rc -ne 0 ]; then echo "!! counter pass $tag failed (rc $rc): no wall.txt is written from partial data"; exit $rc; fi; }
export REPS
run a SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run b SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD
run c MfmaUtil
run d TA_BUSY_avr TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum
python3 - <<'PY'
import csv, glob, collections
out = ["persistent 256 x 256 GEMM (gemm_stream), M = 50432, N = 2304, fp16 — launch kinds of tools/wall_micro.py; counters averaged per launch", ""]
out += open("gpurun_out/pmc_wall/unprofiled.txt").read().strip().splitlines()[-3:] + [""]
kinds = ["A bare K=3072 (no epilogue arithmetic, no stores)", "B shipped K=3072", "C shipped K=768"]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for tag in "abcd":
    per = collections.defaultdict(dict)
    for f in glob.glob(f"gpurun_out/pmc_wall/{tag}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "gemm_stream" not in row["Kernel_Name"]:
                continue
            per[int(row["Dispatch_Id"])][row["Counter_Name"]] = float(row["Counter_Value"])
    ids = sorted(per)                 # REPS + 1 launches per kind (1 warm-up + REPS), in order A, B, C
    import os
    reps = int(os.environ.get("REPS", "2"))
    # the kinds are told apart by dispatch order only: refuse anything but exactly 3 x (REPS + 1) gemm_stream launches (a tail
    # split, a failed pass or another TLXMI_TILE would silently shift the counters onto the wrong kind)
    assert len(ids) == 3 * (reps + 1), f"pass {tag}: {len(ids)} gemm_stream dispatches, expected {3 * (reps + 1)}"
    n = len(ids) // 3
    for i, d in enumerate(ids):
        for c, v in per[d].items():
            vals[kinds[min(i // max(n, 1), 2)]][c].append(v)
for k in kinds:
    a = {c: sum(v) / len(v) for c, v in vals[k].items()}
    out.append(k)
    out.append("  " + "  ".join(f"{c}={a[c]:.4g}" for c in sorted(a)))
    gui = a.get("GRBM_GUI_ACTIVE", 0.0)
    if gui:
        import torch
        prop = torch.cuda.get_device_properties(0) if torch.cuda.is_available() else None
        cus = float(prop.multi_processor_count) if prop else 256.0
        xcds = 8.0 if cus == 256.0 else max(1.0, cus // 32)      # MI355X: 8 XCDs of 32 CUs
        clk = gui / xcds              # GRBM_GUI_ACTIVE is summed over the XCDs
        out.append(f"  kernel cycles (GRBM_GUI_ACTIVE / 8) = {clk:.4g}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in a:
            out.append(f"  MFMA pipe busy     = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x CUs x 4 SIMDs) = {a['SQ_VALU_MFMA_BUSY_CYCLES'] / (clk * cus * 4):.3f}")
        if "SQ_LDS_IDX_ACTIVE" in a:
            out.append(f"  LDS array busy     = SQ_LDS_IDX_ACTIVE / (cycles x CUs)                 = {a['SQ_LDS_IDX_ACTIVE'] / (clk * cus):.3f}")
        for c, nm in (("SQ_ACTIVE_INST_LDS", "LDS issue"), ("SQ_ACTIVE_INST_VMEM", "VMEM issue"), ("SQ_ACTIVE_INST_VALU", "VALU issue"), ("SQ_WAIT_INST_LDS", "LDS issue stall")):
            if c in a:
                out.append(f"  {nm:18s} = 4 x {c} / (cycles x CUs x 4 SIMDs) = {4 * a[c] / (clk * cus * 4):.3f}")
    out.append("")
open("gpurun_out/pmc_wall/wall.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
