"""Swin-B stage-1 Linear layers (K = 128; batch 128 and the half batch of the two-stream forward), hipGraph replay: the
filter-in-registers streaming kernel (gemm_wreg.hip) vs the tiled kernels (TLXMI_WREG=0, tuning flavour)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlxcv_amd import engine as E, _lib
from gconv_micro import timeit
dev = torch.device("cuda:0")
for M in (401408, 200704):
    for name, N, act, res in (("qkv", 384, E.ACT_NONE, False), ("proj", 128, E.ACT_NONE, False), ("fc1 + GELU", 512, E.ACT_GELU, False),
                              ("128 -> 256", 256, E.ACT_NONE, False), ("128 -> 128 + residual", 128, E.ACT_NONE, True)):
        x = torch.randn((M, 128), device=dev).half()
        pk = E.PackedFilter((torch.randn((N, 128, 1, 1), device=dev) * 128 ** -0.5), torch.float16)
        b = torch.randn(N, device=dev) * 0.1
        r = torch.randn((M, N), device=dev).half() if res else None
        f = lambda: E.linear(x, pk, b, r, act)      # noqa: E731
        byt = (M * 128 + M * N * (2 if res else 1)) * 2
        out = [f"tiled {timeit(f, dict(TLXMI_WREG=0)):7.1f}"]
        for wgs in (1, 2, 3):
            out.append(f"streaming, {wgs} workgroups / CU {timeit(f, dict(TLXMI_WREG_WGS=wgs)):7.1f}")
        print(f"M {M} {name:22s} ({byt / 1e6:4.0f} MB = {byt / 5e6:5.1f} us at 5 TB/s) us: " + "   ".join(out), flush=True)
