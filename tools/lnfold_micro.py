"""Folded LayerNorm pieces one by one: time per launch (events around 20 back-to-back launches) and max error against torch, ViT-B/16 and
Swin-B stage-3 shapes at half batch.  usage: lnfold_micro.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import engine as E

dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
torch.manual_seed(0)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000


for (M, K, D, N2, act) in [(25216, 768, 768, 2304, E.ACT_NONE), (25216, 3072, 768, 3072, E.ACT_GELU), (12544, 2048, 512, 1536, E.ACT_NONE), (12544, 2048, 512, 2048, E.ACT_GELU)]:
    x = torch.randn(M, K, device=dev).half()
    w = (torch.randn(D, K, device=dev) / K ** 0.5).half().float()
    b = torch.randn(D, device=dev) * 0.2
    r = torch.randn(M, D, device=dev).half()
    g, be = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev) * 0.3
    w2 = torch.randn(N2, D, device=dev) / D ** 0.5
    b2 = torch.randn(N2, device=dev) * 0.2
    pk = E.PackedFilter(w, torch.float16)
    pk2 = E.PackedFilter(w2, torch.float16)
    prep = E.LinearLN(w2, b2, g, be, torch.float16)
    with E.shared_plan("half"):
        y, part = E.linear_stats(x, pk, b, res=r)
        z = E.linear_ln(y, prep, part, 1e-6, act)
        yref = x.float() @ w.t() + b + r.float()
        e_y = (y.float() - yref).abs().max().item()
        sl = yref.view(M, D // 256, 256)
        e_s = (part[:, :D // 256, 0] - sl.sum(-1)).abs().max().item()
        e_q = (part[:, :D // 256, 1] - (sl * sl).sum(-1)).abs().max().item()
        yf = y.float()
        zref = torch.nn.functional.layer_norm(yf, (D,), g, be, 1e-6) @ w2.t() + b2
        if act == E.ACT_GELU:
            zref = torch.nn.functional.gelu(zref)
        e_z = (z.float() - zref).abs().max().item()
        print(f"M={M} K={K} D={D} N2={N2}: err y {e_y:.2e}  sum {e_s:.2e}  sumsq {e_q:.2e}  consumer {e_z:.2e}", flush=True)
        ln = lambda: E.layernorm(y, g, be, 1e-6)
        yn = ln()
        t = {
            "linear+res": timed(lambda: E.linear(x, pk, b, res=r)),
            "linear_stats+res": timed(lambda: E.linear_stats(x, pk, b, res=r)),
            "layernorm": timed(ln),
            "linear (consumer shape)": timed(lambda: E.linear(yn, pk2, b2, act=act)),
            "linear_ln": timed(lambda: E.linear_ln(y, prep, part, 1e-6, act)),
        }
        print("   " + "   ".join(f"{k} {v:.1f} us" for k, v in t.items()), flush=True)
