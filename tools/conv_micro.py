"""Run one implicit-GEMM conv shape repeatedly (for rocprofv3 --pmc / --kernel-trace passes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tlxcv_amd import engine as E  # noqa: E402

# name: (N, H, W, Cin, Cout, k, stride, res)
SHAPES = {
    "expand56": (256, 56, 56, 64, 256, 1, 1, True),
    "reduce56": (256, 56, 56, 256, 64, 1, 1, False),
    "c3x3_14": (256, 14, 14, 256, 256, 3, 1, False),
    "c3x3_56": (256, 56, 56, 64, 64, 3, 1, False),
    "expand14": (256, 14, 14, 256, 1024, 1, 1, True),
    "expand28": (256, 28, 28, 128, 512, 1, 1, True),
    "expand7": (256, 7, 7, 512, 2048, 1, 1, True),
    "reduce14": (256, 14, 14, 1024, 256, 1, 1, False),
    "c3x3_28": (256, 28, 28, 128, 128, 3, 1, False),
    "c3x3_7": (256, 7, 7, 512, 512, 3, 1, False),
    "trans56": (256, 56, 56, 64, 256, 1, 1, False),
    "qkv": (50432, 1, 1, 768, 2304, 1, 1, False),
    "proj": (50432, 1, 1, 768, 768, 1, 1, True),
    "fc1": (50432, 1, 1, 768, 3072, 1, 1, False),
    "fc2": (50432, 1, 1, 3072, 768, 1, 1, True),
}
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
