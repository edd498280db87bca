"""Which bottleneck seams pay at which batch: ResNet-50 graph replay with all seams / without the 14x14 ones / without any,
interleaved in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tlxcv_amd
from tlxcv_amd import seeded, models, engine as E
from tlxcv_amd.graph import GraphedForward
dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
m = models.resnet50(); m.load_dict(seeded.fill(seeded.shapes_of(m), 1)); m = m.to(dev).set_eval()
orig = E.bottleneck_seam_supported
CFG = {"all seams": (1, orig), "no 14x14 seams": (0, orig), "no seams": (0, lambda *a, **k: False)}
for bs in [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,4,8,16,32,64,96,128,256".split(","))]:
    x = torch.from_numpy(seeded.image_batch(min(bs, 16), 0)).to(dev).repeat((bs + 15) // 16, 1, 1, 1)[:bs].contiguous()
    graphs = {}
    for name, (s256, fn) in CFG.items():
        E.set_option("seam256", s256)
        E.bottleneck_seam_supported = fn
        graphs[name] = GraphedForward(m, x)
    E.bottleneck_seam_supported = orig
    E.set_option("seam256", 1)
    ts = {k: [] for k in CFG}
    for r in range(7):
        for k, f in graphs.items():
            f(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20): f()
            torch.cuda.synchronize()
            ts[k].append(1e3 * (time.perf_counter() - t0) / 20)
    print(f"batch {bs:4d}: " + "   ".join(f"{k}: {sorted(v)[3]:.3f} ms" for k, v in ts.items()), flush=True)
    del graphs
