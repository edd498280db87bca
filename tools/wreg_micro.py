"""Swin-B stage-1 / stage-2 Linear layers (K = 128 / 256 = argv[1]; batch 128 and the half batch of the two-stream forward), hipGraph replay: the
filter-in-registers streaming kernel (gemm_wreg.hip) vs the tiled kernels (TLXMI_WREG=0, tuning flavour)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlxcv_amd import engine as E, _lib
from gconv_micro import timeit
dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 128
CASES = {128: (("qkv", 384, E.ACT_NONE, False), ("proj", 128, E.ACT_NONE, False), ("fc1 + GELU", 512, E.ACT_GELU, False),
               ("128 -> 256", 256, E.ACT_NONE, False), ("128 -> 128 + residual", 128, E.ACT_NONE, True)),
         256: (("qkv", 768, E.ACT_NONE, False), ("proj", 256, E.ACT_NONE, False), ("fc1 + GELU", 1024, E.ACT_GELU, False),
               ("256 -> 512", 512, E.ACT_NONE, False))}[K]
for M in ((401408, 200704) if K == 128 else (100352, 50176)):
    for name, N, act, res in CASES:
        x = torch.randn((M, K), device=dev).half()
        pk = E.PackedFilter((torch.randn((N, K, 1, 1), device=dev) * K ** -0.5), torch.float16)
        b = torch.randn(N, device=dev) * 0.1
        r = torch.randn((M, N), device=dev).half() if res else None
        f = lambda: E.linear(x, pk, b, r, act)      # noqa: E731
        byt = (M * K + M * N * (2 if res else 1)) * 2
        out = [f"tiled {timeit(f, dict(TLXMI_WREG=0)):7.1f}"]
        for wgs in (1, 2, 3):
            out.append(f"streaming, {wgs} workgroups / CU {timeit(f, dict(TLXMI_WREG_WGS=wgs)):7.1f}")
        print(f"M {M} {name:22s} ({byt / 1e6:4.0f} MB = {byt / 5e6:5.1f} us at 5 TB/s) us: " + "   ".join(out), flush=True)
