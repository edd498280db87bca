import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlxcv_amd import engine as E, _lib
_lib.tuning().__enter__()
dev = torch.device("cuda:0")
for rows, C in ((50432, 768), (401408, 128), (100352, 256), (25088, 512), (12544, 512), (6272, 1024), (3136, 1024)):
    x = torch.randn((rows, C), device=dev).half()
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    for _ in range(5): E.layernorm(x, g, b, 1e-6)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20): E.layernorm(x, g, b, 1e-6)
    t1.record(); torch.cuda.synchronize()
    us = 1e3 * t0.elapsed_time(t1) / 20
    print(f"LN {rows}x{C}: {us:.1f} us  {2 * rows * C * 2 / us / 1e3:.0f} GB/s")

# Swin's fused LayerNorm + window plumbing (batch 128, stage 1 / 2)
for B, H, Cc in ((128, 56, 128), (128, 28, 256), (128, 14, 512), (64, 14, 512)):
    x = torch.randn((B, H, H, Cc), device=dev).half()
    g = torch.ones(Cc, device=dev); b = torch.zeros(Cc, device=dev)
    for name, fn in (("LN+partition", lambda: E.layernorm_window_partition(x, g, b, 1e-5, 7, 3)),
                     ("reverse+res+LN", lambda: E.window_reverse_layernorm(x.view(-1, 49, Cc), x, g, b, 1e-5, 7, 3))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(20): fn()
        t1.record(); torch.cuda.synchronize()
        print(f"{name} {B}x{H}x{H}x{Cc}: {1e3 * t0.elapsed_time(t1) / 20:.1f} us")
