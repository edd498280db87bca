"""Experiment: two 4-wave workgroups per CU, the odd one starting `skew` x 512 clocks late (TLXMI_SEAM=16, TLXMI_SEAM_DBG = skew << 8)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import _lib
_lib.tuning().__enter__()
from tlxcv_amd import engine as E

dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
B = 256
SK = [0, 1, 2, 3, 4, 6, 8, 12, 16]
for K1, N1, N2, hw in ((64, 256, 64, 56), (128, 512, 128, 28)):
    g = torch.Generator().manual_seed(1)
    t2 = torch.randn((B, hw, hw, K1), generator=g).half().to(dev)
    skip = torch.randn((B, hw, hw, N1), generator=g).half().to(dev)
    pk3 = E.PackedFilter((torch.randn((N1, K1, 1, 1), generator=g) * (2 / K1) ** 0.5).to(dev), torch.float16)
    pk1 = E.PackedFilter((torch.randn((N2, N1, 1, 1), generator=g) * (2 / N1) ** 0.5).to(dev), torch.float16)
    s3, h3, s1, h1 = (torch.rand(n, generator=g).to(dev) for n in (N1, N1, N2, N2))
    for var in ("16", "0"):
        os.environ["TLXMI_SEAM"] = var
        res = {c: [] for c in SK}
        for r in range(6):
            for c in SK:
                os.environ["TLXMI_SEAM_DBG"] = str(c << 8)
                E.bottleneck_seam(t2, pk3, s3, h3, skip, pk1, s1, h1)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    E.bottleneck_seam(t2, pk3, s3, h3, skip, pk1, s1, h1)
                e1.record()
                torch.cuda.synchronize()
                res[c].append(e0.elapsed_time(e1) / 5 * 1e3)
        print(f"{K1} -> {N1} -> {N2} @ {hw} variant {var}: " + "  ".join(f"[{c}] {sorted(v)[len(v) // 2]:.1f}" for c, v in res.items()), flush=True)
