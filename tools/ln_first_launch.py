"""The fused LayerNorm + Linear (in-kernel statistics): is the FIRST launch of the kernel in a process the odd one, and is it wrong?"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from tlxcv_amd import engine as E
dev = torch.device("cuda:0")
now = E._lib.load()
M, D, cout = 13199, 768, int(sys.argv[1]) if len(sys.argv) > 1 else 3072
warm = len(sys.argv) > 2
g = torch.Generator().manual_seed(0)
x = (torch.randn((M, D), generator=g) * 1.2 + 0.1).half().to(dev)
gamma, beta = (torch.rand(D, generator=g) + 0.5).to(dev), (torch.randn(D, generator=g) * 0.1).to(dev)
w = (torch.randn((cout, D), generator=g) * D ** -0.5).to(dev)
b = (torch.randn(cout, generator=g) * 0.1).to(dev)
prep = E.LinearLN(w, b, gamma, beta, torch.float16)
vp = C.c_void_p


def run(xx, rows):
    y = torch.empty((rows, cout), dtype=torch.float16, device=dev)
    rc = now.tlxmi_layernorm_linear(0, rows, D, cout, D, cout, vp(xx.data_ptr()), vp(prep.pk.buf.data_ptr()), vp(prep.c1.data_ptr()),
                                    vp(prep.c2.data_ptr()), C.c_float(1e-6), 0, vp(y.data_ptr()), None)
    assert rc == 0
    torch.cuda.synchronize()
    return y


if warm:
    run(x[:512].contiguous(), 512)        # a small launch first: code object loaded, instruction cache warm
ys = [run(x, M) for _ in range(4)]
want = F.linear(F.layer_norm(x.float(), (D,), gamma, beta, 1e-6), w, b)
for i, y in enumerate(ys):
    err = (y.float() - want).abs()
    bad = err > 0.02
    print(f"{'warm ' if warm else ''}run {i}: differs from run 1 in {int((y != ys[1]).sum())} elements; {int(bad.sum())} elements off by > 0.02 from the fp32 reference"
          + (f" rows {bad.any(1).nonzero().flatten()[:8].tolist()} cols {bad.any(0).nonzero().flatten()[:8].tolist()}" if bad.any() else ""), flush=True)
