"""Per-launch table of the implicit-GEMM kernel for one ResNet-50 / ViT forward (HIP events)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import tlxcv_amd  # noqa: E402
from tlxcv_amd import engine as E, seeded, models  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
ctor = {"vit_b16": "vit_base_patch16_224", "swin_b": "swintransformer_base_patch4_window7_224"}.get(wl, wl)
m = getattr(models, ctor)()
m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
m = m.to(dev).set_eval()
x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(bs // 32, 1, 1, 1).contiguous()
for _ in range(3):
    m(x)
torch.cuda.synchronize()
reps = 5
probes = []
for _ in range(reps):
    p = []
    E.set_probe(p)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    m(x)
    t1.record()
    torch.cuda.synchronize()
    E.set_probe(None)
    probes.append((p, t0.elapsed_time(t1)))
n = len(probes[0][0])
print(f"{'N,H,W,Cin,Cout,k,s,res':>34} {'us':>8} {'GB/s':>8} {'TF/s':>8}")
tot = 0
for i in range(n):
    us = sorted(1e3 * pr[0][i][0].elapsed_time(pr[0][i][1]) for pr in probes)[reps // 2]
    _, _, b, f, shp = probes[0][0][i]
    tot += us
    print(f"{str(shp):>34} {us:8.1f} {b / us / 1e3:8.0f} {f / us / 1e6:8.1f}")
print(f"sum of gemm launches {tot / 1e3:.3f} ms; whole forward {sorted(p[1] for p in probes)[reps // 2]:.3f} ms")
