"""Does the last partly filled round of the persistent GEMMs cost what the tile arithmetic says?  ViT-B/16 forward (hipGraph replay,
product library) at batch sizes whose half batches give whole rounds of 256 x 256 tiles on 128 CUs for every Linear (220 images:
85 row tiles per half -> qkv 765, proj / fc2 255, fc1 1020 tiles) and at the bench batch 256 (99 row tiles: 891 / 297 / 1188):
images per second, interleaved."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import seeded, models
dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
wl = sys.argv[2] if len(sys.argv) > 2 else "vit_base_patch16_224"
m = getattr(models, wl)()
m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
m = m.to(dev).set_eval()
base = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(20, 1, 1, 1).contiguous()
sizes = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "256,220,218,222,128,110".split(","))]
graphs = {}
for bs in sizes:
    x = base[:bs].contiguous()
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = m(x)
    graphs[bs] = (g, x, y)
ts = {bs: [] for bs in sizes}
for rep in range(7):
    for bs in sizes:
        g = graphs[bs][0]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts[bs].append(e0.elapsed_time(e1) / 10)
for bs, t in ts.items():
    med = sorted(t)[len(t) // 2]
    print(f"batch {bs:4d}: {med:7.3f} ms  {bs / med:7.2f} k img/s", flush=True)
