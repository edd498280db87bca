"""Fixed cost of small launches under hipGraph replay: plain LayerNorm at C = 512 for growing row counts, next to a torch copy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlxcv_amd import engine as E
dev = torch.device("cuda:0")
C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)


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


for rows in (256, 1024, 3136, 12544, 50176, 200704):
    x = torch.randn((rows, C), device=dev).half()
    y = torch.empty_like(x)
    t_ln = timeit(lambda: E.layernorm(x, gam, bet, 1e-5))
    t_cp = timeit(lambda: y.copy_(x))
    mb = 2 * x.numel() * 2 / 1e6
    print(f"rows {rows:7d} ({mb:7.1f} MB in + out): layernorm {t_ln:6.1f} us ({mb / t_ln / 1e3:.2f} TB/s)   torch copy {t_cp:6.1f} us ({mb / t_cp / 1e3:.2f} TB/s)", flush=True)
