"""Experiment: ViT-B/16 batch 256 as two UNEQUAL parts on two streams — the first sized so that its GEMMs fill whole rounds of
256 x 256 tiles (170 row tiles: 510 / 1530 / 2040 tiles for N = 768 / 2304 / 3072), the rest on the side stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tlxcv_amd
from tlxcv_amd import seeded, models, engine as E
dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
wl = sys.argv[1] if len(sys.argv) > 1 else "vit_base_patch16_224"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
splits = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else "0,128,208,216,220,224,232".split(","))]
m = getattr(models, wl)(); m.load_dict(seeded.fill(seeded.shapes_of(m), 1)); m = m.to(dev).set_eval()
x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(bs // 32, 1, 1, 1).contiguous()
E.set_option("two_streams", False)
side = torch.cuda.Stream()


def run(n):
    if n == 0:
        return m(x)
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    y0 = m(x[:n])
    with torch.cuda.stream(side):
        y1 = m(x[n:])
    cur.wait_stream(side)
    return torch.cat((y0, y1), 0)


graphs = {}
for n in splits:
    for _ in range(3):
        run(n)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = run(n)
    graphs[n] = g
ts = {n: [] for n in splits}
for r in range(5):
    for n, g in graphs.items():
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        ts[n].append(1e3 * (time.perf_counter() - t0) / 10)
print(f"{wl} batch {bs}: " + "   ".join(f"{n}+{bs - n}: {sorted(v)[2]:.3f} ms" for n, v in ts.items()), flush=True)
