"""Swin-B window attention (49 tokens, head dim 32) per stage at half batch 64: with the pre-summed bias + mask table vs without any
table (upper bound of what keeping the table out of the per-item traffic could give)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlxcv_amd import engine as E
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator().manual_seed(0)
for stage, (res, heads) in enumerate(((56, 4), (28, 8), (14, 16), (7, 32))):
    nW = (res // 7) ** 2
    qkv = torch.randn((B * nW, 49, 3 * heads * 32), generator=g).half().to(dev)
    bias = torch.randn((heads, 49, 49), generator=g).to(dev)
    mask = (torch.randint(0, 3, (nW, 49), generator=g).unsqueeze(1) != torch.randint(0, 3, (nW, 49), generator=g).unsqueeze(2)).float().to(dev) * -100.0
    tab_m = E.attention_table(bias, mask, 49)
    tab_b = E.attention_table(bias, None, 49)
    cases = {"bias+mask table": lambda: E.attention_comb(qkv, heads, 32 ** -0.5, tab_m, nW),
             "bias table": lambda: E.attention_comb(qkv, heads, 32 ** -0.5, tab_b, 0),
             "no table": lambda: E.attention(qkv, heads, 32 ** -0.5)}
    out = []
    for k, f in cases.items():
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record()
            torch.cuda.synchronize()
            ts.append(100 * e0.elapsed_time(e1))
        out.append(f"{k}: {sorted(ts)[2]:.1f} us")
    byt = qkv.numel() * 2 * 4 / 3
    print(f"stage {stage + 1} ({B * nW * heads} items, {byt / 1e6:.0f} MB of q,k,v,out = {byt / 5e6:.1f} us at 5 TB/s): " + "   ".join(out), flush=True)
