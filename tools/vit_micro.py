"""ViT-B/16 block pieces at batch 256 (M = 50432), fp16, interleaved in one process (HIP events, median of rounds):
LayerNorm, qkv, fc1, proj, fc2, attention."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import tlxcv_amd  # noqa: E402
from tlxcv_amd import engine as E  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M, D = B * 197, 768
g = torch.Generator().manual_seed(0)
x = (torch.randn((M, D), generator=g) * 1.2 + 0.1).half().to(dev)
res = x.clone()
h = torch.randn((M, 4 * D), generator=g).half().to(dev)
gamma, beta = (torch.rand(D, generator=g) + 0.5).to(dev), (torch.randn(D, generator=g) * 0.1).to(dev)


def lin(cin, cout):
    w = (torch.randn((cout, cin), generator=g) * cin ** -0.5).to(dev)
    b = (torch.randn(cout, generator=g) * 0.1).to(dev)
    return w, b, E.PackedFilter(w, torch.float16), None


wq, bq, pkq, lnq = lin(D, 3 * D)
w1, b1, pk1, ln1 = lin(D, 4 * D)
wp, bp, pkp, _ = lin(D, D)
w2, b2, pk2, _ = lin(4 * D, D)
qkv = torch.randn((B, 197, 3 * D), generator=g).half().to(dev)

cases = {
    "layernorm": lambda: E.layernorm(x, gamma, beta, 1e-6),
    "qkv plain": lambda: E.linear(x, pkq, bq),
    "fc1 plain (gelu)": lambda: E.linear(x, pk1, b1, act=E.ACT_GELU),
    "proj + res": lambda: E.linear(x, pkp, bp, res=res, out=res),
    "fc2 + res": lambda: E.linear(h, pk2, b2, res=res, out=res),
    "attention": lambda: E.attention(qkv, 12, 0.125),
}
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
if only:
    cases = {k: v for k, v in cases.items() if any(o in k for o in only)}
for f in cases.values():
    f()
torch.cuda.synchronize()
rounds, inner = 7, 6
times = {k: [] for k in cases}
for _ in range(rounds):
    for k, f in cases.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            f()
        e1.record()
        torch.cuda.synchronize()
        times[k].append(1e3 * e0.elapsed_time(e1) / inner)
flops = {"qkv": 2 * M * D * 3 * D, "fc1": 2 * M * D * 4 * D, "proj": 2 * M * D * D, "fc2": 2 * M * D * 4 * D}
for k, v in times.items():
    v.sort()
    med = v[len(v) // 2]
    fl = next((f for n, f in flops.items() if k.startswith(n)), 0)
    print(f"{k:22s} {med:8.1f} us   (min {v[0]:7.1f})" + (f"   {fl / med / 1e6:7.0f} TF/s" if fl else ""), flush=True)
