"""Calibration only: time torch.matmul (hipBLASLt/rocBLAS) on the ViT GEMM shapes."""
import torch
for nm, (M, K, N) in {"qkv": (50432, 768, 2304), "proj": (50432, 768, 768), "fc1": (50432, 768, 3072), "fc2": (50432, 3072, 768)}.items():
    x = torch.randn(M, K, device="cuda", dtype=torch.half)
    w = torch.randn(N, K, device="cuda", dtype=torch.half)
    for _ in range(5):
        y = x @ w.t()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        y = x @ w.t()
    t1.record()
    torch.cuda.synchronize()
    us = 1e3 * t0.elapsed_time(t1) / 20
    print(f"{nm}: torch.matmul {us:.1f} us  {2.0 * M * K * N / us / 1e6:.0f} TFLOP/s")
