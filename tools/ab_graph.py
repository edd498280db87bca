"""A/B of a tuning variable on the hipGraph replay of a whole forward (tuning flavour; the variable is read at capture time):
one graph per setting, replays interleaved.  usage: ab_graph.py VAR v1,v2,... [ctor=resnet50] [batch=256]   ("-" = unset)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import seeded, models, engine as E, _lib
_lib.tuning().__enter__()

var, vals = sys.argv[1], (sys.argv[2].split(";") if ";" in sys.argv[2] else sys.argv[2].split(","))
wl = sys.argv[3] if len(sys.argv) > 3 else "resnet50"
bs = int(sys.argv[4]) if len(sys.argv) > 4 else 256
dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
ctor = {"vit_b16": "vit_base_patch16_224", "swin_b": "swintransformer_base_patch4_window7_224"}.get(wl, wl)
m = getattr(models, ctor)()
m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
m = m.to(dev).set_eval()
x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(bs // 32, 1, 1, 1).contiguous()


def setv(v):
    if var.startswith("opt:"):
        E.set_option(var[4:], int(v))
    elif v == "-":
        os.environ.pop(var, None)
    else:
        os.environ[var] = v


graphs = {}
for v in vals:
    setv(v)
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = m(x)
    graphs[v] = (g, y)
    g.replay()
torch.cuda.synchronize()
ts = {v: [] for v in vals}
for rep in range(7):
    for v in vals:
        g = graphs[v][0]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts[v].append(e0.elapsed_time(e1) / 10)
print(f"{wl} batch {bs}  " + "   ".join(f"{var}={v}: {sorted(t)[len(t) // 2]:.3f} ms" for v, t in ts.items()), flush=True)
torch.cuda.synchronize()
