import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, tlxcv_amd
from tlxcv_amd import seeded, models, engine as E
dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
wl, bs = sys.argv[1], int(sys.argv[2])
m = getattr(models, wl)(); m.load_dict(seeded.fill(seeded.shapes_of(m), 1)); m = m.to(dev).set_eval()
x = torch.from_numpy(seeded.image_batch(16, 0)).to(dev).repeat(bs // 16, 1, 1, 1).contiguous()
E.set_option("two_streams", False)
cfgs = {"one stream": lambda: m(x), "halves": lambda: E.run_halves(m, x, None), "halves/full": lambda: E.run_halves(m, x, "full"), "halves/half": lambda: E.run_halves(m, x, "half")}
graphs = {}
for k, f in cfgs.items():
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = f()
    graphs[k] = g
ts = {k: [] for k in cfgs}
for r in range(5):
    for k, g in graphs.items():
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        ts[k].append(1e3 * (time.perf_counter() - t0) / 10)
print(f"{wl} batch {bs}: " + "   ".join(f"{k}: {sorted(v)[2]:.3f} ms" for k, v in ts.items()), flush=True)
