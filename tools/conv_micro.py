"""Run one implicit-GEMM conv shape repeatedly (for rocprofv3 --pmc / --kernel-trace passes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tlxcv_amd import engine as E  # noqa: E402

from tools.conv_micro_shapes import SHAPES  # noqa: E402

names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(SHAPES)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for nm in names:
    if nm.startswith("attn"):
        B, Nt, heads, hd = (256, 197, 12, 64) if nm == "attn_vit" else (8192, 49, 4, 32)
        qkv = (torch.randn((B, Nt, 3 * heads * hd), generator=g) * 0.5).half().to(dev)
        bias = torch.randn((heads, Nt, Nt), generator=g).to(dev) if nm == "attn_swin" else None
        for _ in range(reps):
            y = E.attention(qkv, heads, hd ** -0.5, bias)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(reps):
            y = E.attention(qkv, heads, hd ** -0.5, bias)
        t1.record()
        torch.cuda.synchronize()
        print(f"{nm}: {1e3 * t0.elapsed_time(t1) / reps:.1f} us")
        continue
    N, H, W, Ci, Co, k, st, res = SHAPES[nm]
    x = (torch.randn((N, H, W, Ci), generator=g) * 0.5).half().to(dev)
    w = torch.randn((Co, Ci, k, k), generator=g) * (2.0 / (Ci * k * k)) ** 0.5
    pk = E.PackedFilter(w.to(dev), torch.float16)
    sc = torch.ones(Co, device=dev)
    sh = torch.zeros(Co, device=dev)
    Ho = (H + 2 * (k // 2) - k) // st + 1
    r = (torch.randn((N, Ho, Ho if H > 1 else 1, Co), generator=g)).half().to(dev) if res else None
    act = E.ACT_GELU if nm == "fc1" else E.ACT_RELU
    for _ in range(reps):
        y = E.conv2d(x, pk, st, k // 2, 1, sc, sh, r, act)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        y = E.conv2d(x, pk, st, k // 2, 1, sc, sh, r, act)
    t1.record()
    torch.cuda.synchronize()
    print(f"{nm}: {1e3 * t0.elapsed_time(t1) / reps:.1f} us")
