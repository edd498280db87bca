"""Which fused LayerNorm + Linear launch is not bit-reproducible at ViT-B/16's full size, and where."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlxcv_amd import engine as E
dev = torch.device("cuda:0")
M, D = 256 * 197, 768
g = torch.Generator().manual_seed(0)
x = (torch.randn((M, D), generator=g) * 1.2 + 0.1).half().to(dev)
gamma, beta = (torch.rand(D, generator=g) + 0.5).to(dev), (torch.randn(D, generator=g) * 0.1).to(dev)
for name, cout, act in (("qkv", 2304, E.ACT_NONE), ("fc1", 3072, E.ACT_GELU)):
    w = (torch.randn((cout, D), generator=g) * D ** -0.5).to(dev)
    b = (torch.randn(cout, generator=g) * 0.1).to(dev)
    prep = E.LinearLN(w, b, gamma, beta, torch.float16)
    ref = E.linear_ln(x, prep, 1e-6, act, in_kernel=True).clone()
    torch.cuda.synchronize()
    for it in range(8):
        y = E.linear_ln(x, prep, 1e-6, act, in_kernel=True)
        torch.cuda.synchronize()
        d = (y != ref)
        n = int(d.sum())
        if n:
            rows = d.any(1).nonzero().flatten()
            cols = d.any(0).nonzero().flatten()
            print(f"{name} run {it}: {n} elements differ; rows {len(rows)}: {rows[:12].tolist()} .. {rows[-3:].tolist()}; cols {len(cols)}: {cols[:8].tolist()} .. {cols[-3:].tolist()}; max |d| {float((y.float() - ref.float()).abs().max()):.4g}")
            r = rows[0].item()
            dc = d[r].nonzero().flatten()
            print(f"   row {r}: {len(dc)} cols differ, first {dc[:6].tolist()}, row % 256 = {r % 256}, tile row block {(r % 256) // 16}")
        else:
            print(f"{name} run {it}: identical")
