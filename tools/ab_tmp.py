import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import seeded, models, engine as E
dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
m = models.swintransformer_base_patch4_window7_224()
m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
m = m.to(dev).set_eval()
bs = 128
x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(bs // 32, 1, 1, 1).contiguous()
fused = E.patch_merge_layernorm
def two(x, g, b, eps):
    B, H, W, Cc = x.shape
    return E.layernorm(E.patch_merge_gather(x).view(B, (H // 2) * (W // 2), 4 * Cc), E._f32(g), E._f32(b), eps)
graphs = {}
for name, fn in (("fused", fused), ("two", two)):
    E.patch_merge_layernorm = fn
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = m(x)
    g.replay(); torch.cuda.synchronize()
    graphs[name] = (g, y.clone())
print("equal outputs:", torch.equal(graphs["fused"][1], graphs["two"][1]))
ts = {k: [] for k in graphs}
for rep in range(9):
    for k, (g, _) in graphs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        ts[k].append(e0.elapsed_time(e1) / 10)
print("  ".join(f"{k}: {sorted(t)[len(t)//2]:.3f} ms" for k, t in ts.items()))
