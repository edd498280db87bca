"""ResNet-50 head at half batch: global average pool (128, 7, 7, 2048) and the 2048 -> 1000 classifier, as dispatched vs K slices."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlxcv_amd import engine as E
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n):
            f()
    gr.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(1e3 * e0.elapsed_time(e1) / n)
    return sorted(ts)[3]


x = torch.randn((B, 7, 7, 2048), device=dev).half()
print(f"global_avgpool ({B},7,7,2048): {timeit(lambda: E.global_avgpool(x)):.1f} us")
f = torch.randn((B, 2048), device=dev).half()
w = (torch.randn((1000, 2048), device=dev) * 0.02)
b = torch.randn(1000, device=dev) * 0.1
pk = E.PackedFilter(w, torch.float16)
print(f"fc 2048 -> 1000, {B} rows, as dispatched: {timeit(lambda: E.linear(f, pk, b)):.1f} us")
orig = E._linear_splits
for s in (2, 4, 8, 16):
    E._linear_splits = lambda rows, K, pk_, x_, s=s: s
    print(f"fc with {s} K slices: {timeit(lambda: E.linear(f, pk, b)):.1f} us")
E._linear_splits = orig
