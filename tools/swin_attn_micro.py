"""Swin-B window attention (49 tokens, head dim 32) per stage at half batch 64 (hipGraph replay): the table-resident kernel
(workgroups bound to one (window position, head pair), table in LDS) vs the streaming form (16 KB of table read per item) vs no
table at all (the floor of what keeping the table out of the per-item traffic can give).  Knob: TLXMI_WIN_WPC (workgroups per CU
the grid is sized for)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlxcv_amd import engine as E, _lib
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator().manual_seed(0)


def timeit(f, env):
    with _lib.tuning(**env):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(10):
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
        ts.append(100 * e0.elapsed_time(e1))
    return sorted(ts)[3]


for stage, (res, heads) in enumerate(((56, 4), (28, 8), (14, 16), (7, 32))):
    nW = (res // 7) ** 2
    qkv = torch.randn((B * nW, 49, 3 * heads * 32), generator=g).half().to(dev)
    bias = torch.randn((heads, 49, 49), generator=g).to(dev)
    mask = (torch.randint(0, 3, (nW, 49), generator=g).unsqueeze(1) != torch.randint(0, 3, (nW, 49), generator=g).unsqueeze(2)).float().to(dev) * -100.0
    tab_m = E.attention_table(bias, mask, 49)
    tab_b = E.attention_table(bias, None, 49)
    fm = lambda: E.attention_comb(qkv, heads, 32 ** -0.5, tab_m, nW)      # noqa: E731
    fb = lambda: E.attention_comb(qkv, heads, 32 ** -0.5, tab_b, 0)       # noqa: E731
    out = [f"no table {timeit(lambda: E.attention(qkv, heads, 32 ** -0.5), {}):.1f}"]
    out.append(f"streamed: mask {timeit(fm, dict(TLXMI_WIN_STREAM=1)):.1f} bias {timeit(fb, dict(TLXMI_WIN_STREAM=1)):.1f}")
    for wpc in (2, 3, 4, 6):
        out.append(f"resident wpc {wpc}: mask {timeit(fm, dict(TLXMI_WIN_WPC=wpc)):.1f} bias {timeit(fb, dict(TLXMI_WIN_WPC=wpc)):.1f}")
    byt = qkv.numel() * 2 * 4 / 3
    print(f"stage {stage + 1} ({B * nW * heads} items, {byt / 1e6:.0f} MB = {byt / 5e6:.1f} us at 5 TB/s) us: " + " | ".join(out), flush=True)
