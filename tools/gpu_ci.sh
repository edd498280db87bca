#!/bin/bash
# One GPU-box visit: smoke -> gpu tests -> bench.  Stops at the first step that is killed/timed out
# (never start another GPU step after a hang).  Logs go to gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
step() {  # name timeout cmd...
    local name=$1 tmo=$2; shift 2
    echo "== $name"
    timeout -k 10 "$tmo" "$@" > "gpurun_out/$name.log" 2>&1
    local rc=$?
    echo "== $name rc=$rc"
    tail -n "${TAILN:-15}" "gpurun_out/$name.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "!! $name timed out/killed: stopping"; exit $rc; fi
    return 0
}
for s in "$@"; do
    case $s in
        smoke) step smoke 400 python -c "import __graft_entry__ as g; g.smoke()" ;;
        tests) step pytest 900 python -m pytest tests -m gpu -q -x --timeout=600 ;;
        testsall) step pytest 900 python -m pytest tests -m gpu -q --timeout=600 ;;
        bench) step bench 600 python bench.py ;;
        benchvit) step benchvit 600 python bench.py --workload vit_b16 --steps 10 --warmup 3 --no-cpu-baseline ;;
        benchswin) step benchswin 600 python bench.py --workload swin_b --steps 10 --warmup 3 --no-cpu-baseline ;;
        prof) cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
              step prof 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline ;;
        *) echo "unknown step $s" ;;
    esac
done
