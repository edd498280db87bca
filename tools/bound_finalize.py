"""Upper bound of folding tlxmi_ln_finalize into its producer or consumer: the hipGraph replay of the whole forward with the finalize
launches dropped (the consumers read a stale row table: results wrong, timing only), interleaved with the product graph.
usage: bound_finalize.py [vit_b16|swin_b] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import seeded, models, engine as E

wl = sys.argv[1] if len(sys.argv) > 1 else "vit_b16"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
ctor = {"vit_b16": "vit_base_patch16_224", "swin_b": "swintransformer_base_patch4_window7_224"}[wl]
m = getattr(models, ctor)()
m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
m = m.to(dev).set_eval()
x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(bs // 32, 1, 1, 1).contiguous()
real = E.ln_finalize
stale = {}


def dropped(part, C, eps):
    k = (part.shape[1], torch.cuda.current_stream().cuda_stream)
    if k not in stale:
        stale[k] = torch.zeros((part.shape[1], 2), dtype=torch.float32, device=part.device)
    return stale[k]


graphs = {}
for name, fn in (("product", real), ("finalize launches dropped", dropped)):
    E.ln_finalize = fn
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = m(x)
    graphs[name] = g
    g.replay()
E.ln_finalize = real
torch.cuda.synchronize()
ts = {k: [] for k in graphs}
for rep in range(7):
    for k, g in graphs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts[k].append(e0.elapsed_time(e1) / 10)
base = sorted(ts["product"])[3]
for k, t in ts.items():
    med = sorted(t)[3]
    print(f"{wl} batch {bs}  {k:28s} {med:7.3f} ms   {100 * (med - base) / base:+6.2f} %", flush=True)
