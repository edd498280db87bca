"""Poisoned-LDS repeat test of the fused LayerNorm + Linear (ROWAFF 2) for one library build (TLXMI_LIB)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlxcv_amd import engine as E
dev = torch.device("cuda:0")
P = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libpoison.so"))
M, D = 13199, 768
g = torch.Generator().manual_seed(0)
x = (torch.randn((M, D), generator=g) * 1.2 + 0.1).half().to(dev)
gamma, beta = (torch.rand(D, generator=g) + 0.5).to(dev), (torch.randn(D, generator=g) * 0.1).to(dev)
vp = C.c_void_p
st = torch.cuda.current_stream().cuda_stream
tot = 0
for cout in (3072, 768):
    w = (torch.randn((cout, D), generator=g) * D ** -0.5).to(dev)
    b = (torch.randn(cout, generator=g) * 0.1).to(dev)
    prep = E.LinearLN(w, b, gamma, beta, torch.float16)
    f = lambda: E.linear_ln(x, prep, 1e-6, E.ACT_NONE, in_kernel=True)
    ref = f().clone()
    torch.cuda.synchronize()
    nd = []
    for k in range(10):
        P.poison_lds(C.c_uint(0x7fc07fc0 if k & 1 else 0x3c003c00), vp(st))
        y = f()
        torch.cuda.synchronize()
        nd.append(int(((y != ref) | torch.isnan(y)).sum()))
    tot += sum(nd)
    print(os.path.basename(os.environ.get("TLXMI_LIB", "product")), f"Cout={cout}: differing elements per poisoned run {nd}", flush=True)
