"""Swin window attention at the four stage shapes: tlxmi_attention (bias + mask) vs tlxmi_attention_comb."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tlxcv_amd import engine as E  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return 1e3 * t0.elapsed_time(t1) / reps


for stage, (Bw, heads, nW) in enumerate([(8192, 4, 64), (2048, 8, 16), (512, 16, 4), (128, 32, 1)], 1):
    N, hd = 49, 32
    qkv = (torch.randn((Bw, N, 3 * heads * hd), generator=g) * 0.5).half().to(dev)
    bias = torch.randn((heads, N, N), generator=g).to(dev)
    mask = (torch.randn((nW, N, N), generator=g) > 1).float().to(dev) * -100.0
    tab = E.attention_table(bias, mask, N)
    a = timeit(lambda: E.attention(qkv, heads, hd ** -0.5, bias, mask))
    b = timeit(lambda: E.attention_comb(qkv, heads, hd ** -0.5, tab, nW))
    mb = (qkv.numel() + Bw * N * heads * hd) * 2 / 1e6
    print(f"stage {stage}: items {Bw * heads:6d}  attention {a:7.1f} us   attention_comb {b:7.1f} us   ({mb:.0f} MB -> {mb / 5e3 * 1e3:.0f} us at 5 TB/s)", flush=True)
