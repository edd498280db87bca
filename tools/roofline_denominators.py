"""Measured roofline denominators on this box (SURVEY §8d asks for spec AND measured): HBM copy / read / write
bandwidth with plain torch kernels, and the fp16 MFMA rate of a large GEMM — hipBLASLt through torch.matmul and this
repository's own 256 x 256 antiphase kernel on the same shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import tlxcv_amd  # noqa: E402,F401
from tlxcv_amd import engine as E  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return 1e-3 * t0.elapsed_time(t1) / n     # seconds


GB = 1 << 30
for gib in (1, 4):
    a = torch.empty(gib * GB // 2, dtype=torch.float16, device=dev).fill_(1.0)
    b = torch.empty_like(a)
    s = timed(lambda: b.copy_(a))
    print(f"copy  {gib} GiB -> {gib} GiB : {2 * gib * GB / s / 1e12:6.2f} TB/s (read + write)")
    s = timed(lambda: a.sum())
    print(f"read  {gib} GiB (sum)      : {gib * GB / s / 1e12:6.2f} TB/s")
    s = timed(lambda: b.fill_(2.0))
    print(f"write {gib} GiB (fill)     : {gib * GB / s / 1e12:6.2f} TB/s")
    del a, b
    torch.cuda.empty_cache()

for M, K, N in ((8192, 8192, 8192), (50432, 768, 2304), (16384, 4096, 4096)):
    x = (torch.randn(M, K, device=dev) * 0.1).half()
    w = (torch.randn(N, K, device=dev) * 0.1).half()
    s = timed(lambda: x @ w.t(), n=10)
    print(f"GEMM {M}x{K}x{N} fp16  torch.matmul (hipBLASLt): {2.0 * M * K * N / s / 1e12:7.1f} TFLOP/s")
    pk = E.PackedFilter(w.float(), torch.float16)
    s = timed(lambda: E.linear(x, pk), n=10)
    print(f"GEMM {M}x{K}x{N} fp16  tlxmi_conv2d (dispatcher)   : {2.0 * M * K * N / s / 1e12:7.1f} TFLOP/s")
    del x, w, pk
    torch.cuda.empty_cache()
