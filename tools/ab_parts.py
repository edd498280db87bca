"""Whole forward as P equal parts on P streams inside one hipGraph (P = 1: the model's own forward, i.e. its two halves) — the
question behind bench.py's two_in_flight figure: does more overlap inside ONE step help?  usage: ab_parts.py ctor batch P1,P2,.. [plan]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import seeded, models, engine as E

wl = sys.argv[1] if len(sys.argv) > 1 else "swin_b"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 128
parts = [int(p) for p in (sys.argv[3] if len(sys.argv) > 3 else "1,4").split(",")]
plan = sys.argv[4] if len(sys.argv) > 4 else "full"
dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
ctor = {"vit_b16": "vit_base_patch16_224", "swin_b": "swintransformer_base_patch4_window7_224"}.get(wl, wl)
m = getattr(models, ctor)()
m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
m = m.to(dev).set_eval()
x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(bs // 32, 1, 1, 1).contiguous()
streams = [torch.cuda.Stream(device=dev) for _ in range(8)]


def fwd(P):
    if P == 1:
        return m(x)
    E.set_option("two_streams", False)
    try:
        cur = torch.cuda.current_stream(dev)
        n = bs // P
        ys = []
        with E.shared_plan(plan):
            for i in range(P):
                s = streams[i]
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    ys.append(m(x[i * n:(i + 1) * n]))
            for i in range(P):
                cur.wait_stream(streams[i])
        return torch.cat(ys, 0)
    finally:
        E.set_option("two_streams", True)


graphs = {}
ref = None
for P in parts:
    for _ in range(3):
        fwd(P)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = fwd(P)
    g.replay()
    torch.cuda.synchronize()
    if ref is None:
        ref = y.float().clone()
    else:
        print(f"P={P} max|diff| vs first {float((y.float() - ref).abs().max()):.3e}")
    graphs[P] = g
ts = {P: [] for P in parts}
for rep in range(7):
    for P in parts:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            graphs[P].replay()
        e1.record()
        torch.cuda.synchronize()
        ts[P].append(e0.elapsed_time(e1) / 10)
print(f"{wl} batch {bs} plan {plan}  " + "   ".join(f"parts={P}: {sorted(t)[len(t) // 2]:.3f} ms" for P, t in ts.items()), flush=True)
