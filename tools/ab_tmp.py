import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tlxcv_amd
from tlxcv_amd import seeded, models, engine as E
dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
m = models.resnet50(); m.load_dict(seeded.fill(seeded.shapes_of(m), 1)); m = m.to(dev).set_eval()
x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(8, 1, 1, 1).contiguous()
orig = E._linear_splits
def lowered(rows, K, pk, x_):
    es = x_.element_size()
    if not E._options["splitk"] or rows > 512 or pk.Cin != K or pk.Cin_pad != K: return 0
    if K * es < 4096 or pk.Cout * K * es < 3000000 or (pk.Cout * es) % 16: return 0
    kt = K * es // 128
    tiles = ((rows + 63) // 64) * ((pk.Cout + 63) // 64)
    best = 0
    for s_ in range(2, 65):
        if kt % s_ or kt // s_ < 4: continue
        best = s_
        if tiles * s_ >= 2 * 256: break
    return best
graphs = {}
for name, fn in {"head one launch": orig, "head on K slices": lowered}.items():
    E._linear_splits = fn
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): y = m(x)
    graphs[name] = g
ts = {k: [] for k in graphs}
for rep in range(9):
    for k, g in graphs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): g.replay()
        e1.record(); torch.cuda.synchronize()
        ts[k].append(e0.elapsed_time(e1) / 10)
print("  ".join(f"{k}: {sorted(v)[4]:.4f} ms" for k, v in ts.items()))
