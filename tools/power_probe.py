"""Is the forward clock / power bound?  Same hipGraph (same launches, same bytes), three data sets: seeded weights + images (the
bench), all-zero weights + images, and constant 1.0 — the MFMA array's power depends on how many operand bits toggle
(MI355X_MICROARCH.md), so a large gap means the chip is holding a power limit, not waiting for a schedule.
usage: power_probe.py [ctor=vit_b16] [batch=256]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tlxcv_amd
from tlxcv_amd import seeded, models, engine as E

wl = sys.argv[1] if len(sys.argv) > 1 else "vit_b16"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
ctor = {"vit_b16": "vit_base_patch16_224", "swin_b": "swintransformer_base_patch4_window7_224"}.get(wl, wl)


def build(kind):
    m = getattr(models, ctor)()
    w = seeded.fill(seeded.shapes_of(m), 1)
    if kind != "seeded":
        c = 0.0 if kind == "zeros" else 1.0
        w = {k: np.full_like(v, c) for k, v in w.items()}
    m.load_dict(w)
    m = m.to(dev).set_eval()
    x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(bs // 32, 1, 1, 1).contiguous()
    if kind != "seeded":
        x = torch.full_like(x, 0.0 if kind == "zeros" else 1.0)
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = m(x)
    g.replay()
    torch.cuda.synchronize()
    return g, m, x, y


graphs = {k: build(k) for k in ("seeded", "zeros", "ones")}
ts = {k: [] for k in graphs}
for rep in range(7):
    for k, (g, *_rest) in graphs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts[k].append(e0.elapsed_time(e1) / 20)
print(f"{wl} batch {bs}  " + "   ".join(f"{k}: {sorted(t)[len(t) // 2]:.3f} ms" for k, t in ts.items()), flush=True)
