"""In-process A/B of a per-call tuning variable (TLXMI_TAIL6, TLXMI_TAIL, TLXMI_HALO, TLXMI_TILE ...): the same model,
the two settings interleaved pass by pass on one box, whole-forward time and per-layer-shape time for each.
usage: python tools/ab_env.py VAR A B [ctor=resnet50] [batch=256] [reps=6]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import tlxcv_amd  # noqa: E402,F401
from tlxcv_amd import seeded, models, engine as E, _lib  # noqa: E402

_lib.tuning().__enter__()      # the whole run goes through libtlxmi_tune.so, the flavour that reads the knobs per call

var, va, vb = sys.argv[1:4]
ctor = sys.argv[4] if len(sys.argv) > 4 else "resnet50"
bs = int(sys.argv[5]) if len(sys.argv) > 5 else 256
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 6
dev = torch.device("cuda:0")
m = getattr(models, ctor)()
m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
m = m.to(dev).set_eval()
x = torch.from_numpy(seeded.image_batch(16, 0)).to(dev).repeat(bs // 16, 1, 1, 1).contiguous()


def setv(v):
    if v == "-":
        os.environ.pop(var, None)
    else:
        os.environ[var] = v


tot = {va: 0.0, vb: 0.0}
per = {va: {}, vb: {}}
for v in (va, vb):
    setv(v)
    for _ in range(2):
        m(x)
torch.cuda.synchronize()
for r in range(reps):
    for v in ((va, vb) if r % 2 == 0 else (vb, va)):
        setv(v)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        tot[v] += (time.perf_counter() - t0) / 3
        probe = []
        E.set_probe(probe)
        m(x)
        torch.cuda.synchronize()
        E.set_probe(None)
        for e0, e1, b, f, shape in probe:
            a = per[v].setdefault(shape, [0, 0.0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1) * 1e3
print(f"{ctor} bs{bs}: {var}={va}: {1e3 * tot[va] / reps:.3f} ms   {var}={vb}: {1e3 * tot[vb] / reps:.3f} ms")
for shape in per[va]:
    ua = per[va][shape][1] / per[va][shape][0]
    ub = per[vb][shape][1] / per[vb][shape][0]
    mark = " <<<" if abs(ua - ub) > 0.03 * ua else ""
    print(f"  {str(shape):50s} x{per[va][shape][0] // reps:2d}  {ua:8.1f}  {ub:8.1f} us{mark}")
