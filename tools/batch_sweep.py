"""ResNet-50 / ViT-B/16 forward time at batch 1 .. 256 (fp16, hipGraph replay and per-kernel launches): the dispatcher off its
tuned point.  Output -> profiles/<round>/batch_sweep.txt."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import tlxcv_amd  # noqa: E402
from tlxcv_amd import seeded, models  # noqa: E402
from tlxcv_amd.graph import GraphedForward  # noqa: E402

dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
for ctor in ("resnet50", "vit_base_patch16_224"):
    m = getattr(models, ctor)()
    m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
    m = m.to(dev).set_eval()
    for bs in (1, 8, 32, 64, 128, 256):
        x = torch.from_numpy(seeded.image_batch(min(bs, 16), 0)).to(dev).repeat((bs + 15) // 16, 1, 1, 1)[:bs].contiguous()
        res = {}
        for mode in ("eager", "graph"):
            f = GraphedForward(m, x) if mode == "graph" else (lambda: m(x))
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            n = 20
            t0 = time.perf_counter()
            for _ in range(n):
                f()
            torch.cuda.synchronize()
            res[mode] = 1e3 * (time.perf_counter() - t0) / n
        print(f"{ctor:22s} batch {bs:4d}: graph {res['graph']:8.3f} ms ({bs / res['graph'] * 1e3:8.0f} img/s)   per-kernel launches {res['eager']:8.3f} ms", flush=True)
    del m
    torch.cuda.empty_cache()
