"""A/B of tile candidates on GEMM / conv shapes, interleaved rounds in ONE process (medians).
usage: python tools/ab_tiles.py qkv,fc1 7,8,-1 [rounds] [reps]      (-1 = the dispatcher's own choice)"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tlxcv_amd import engine as E, _lib  # noqa: E402
_lib.tuning().__enter__()      # the flavour of the library that reads TLXMI_TILE per call
from tools.conv_micro_shapes import SHAPES  # noqa: E402

names = sys.argv[1].split(",")
tiles = [int(v) for v in sys.argv[2].split(",")]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for nm in names:
    N, H, W, Ci, Co, k, st, res = SHAPES[nm]
    x = (torch.randn((N, H, W, Ci), generator=g) * 0.5).half().to(dev)
    w = torch.randn((Co, Ci, k, k), generator=g) * (2.0 / (Ci * k * k)) ** 0.5
    pk = E.PackedFilter(w.to(dev), torch.float16)
    lin = H == 1 and W == 1
    sc = None if lin else torch.ones(Co, device=dev)
    sh = torch.zeros(Co, device=dev)
    Ho = (H + 2 * (k // 2) - k) // st + 1
    r = (torch.randn((N, Ho, Ho if H > 1 else 1, Co), generator=g)).half().to(dev) if res else None
    act = E.ACT_GELU if nm.endswith("fc1") else (E.ACT_NONE if lin else E.ACT_RELU)
    times = {t: [] for t in tiles}
    for rd in range(rounds + 1):
        for t in tiles:
            if t < 0:
                os.environ.pop("TLXMI_TILE", None)
            else:
                os.environ["TLXMI_TILE"] = str(t)
            y = E.conv2d(x, pk, st, k // 2, 1, sc, sh, r, act)
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(reps):
                y = E.conv2d(x, pk, st, k // 2, 1, sc, sh, r, act)
            t1.record()
            torch.cuda.synchronize()
            if rd > 0:
                times[t].append(1e3 * t0.elapsed_time(t1) / reps)
    if lin:   # calibration: the library GEMM alone (no bias / activation / residual), same process
        x2, w2 = x.reshape(N, Ci), w.reshape(Co, Ci).half().to(dev)
        for _ in range(3):
            y2 = x2 @ w2.t()
        torch.cuda.synchronize()
        tm = []
        for rd in range(rounds):
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(reps):
                y2 = x2 @ w2.t()
            t1.record()
            torch.cuda.synchronize()
            tm.append(1e3 * t0.elapsed_time(t1) / reps)
        print(nm, f"torch.matmul {statistics.median(tm):.1f}", end="  ")
    print(nm, "  ".join(f"tile {t}: {statistics.median(v):.1f} (min {min(v):.1f})" for t, v in times.items()), flush=True)
