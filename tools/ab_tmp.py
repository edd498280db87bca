import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tlxcv_amd
from tlxcv_amd import seeded, models, engine as E
dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
m = models.vit_base_patch16_224(); m.load_dict(seeded.fill(seeded.shapes_of(m), 1)); m = m.to(dev).set_eval()
x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(8, 1, 1, 1).contiguous()
graphs = {}
for name, (opt, mink) in {"off": (0, 12), "k>=768": (1, 12), "k>=3072": (1, 48)}.items():
    E.set_option("tail_splitk", opt); E._TAIL_MIN_KTILES = mink
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): y = m(x)
    graphs[name] = g
ts = {k: [] for k in graphs}
for rep in range(7):
    for k, g in graphs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): g.replay()
        e1.record(); torch.cuda.synchronize()
        ts[k].append(e0.elapsed_time(e1) / 10)
print("  ".join(f"{k}: {sorted(v)[3]:.3f} ms" for k, v in ts.items()))
