"""Swin-B LayerNorm-type passes per stage at half batch: norm1 + roll + window partition, and window reverse + roll + residual + norm2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlxcv_amd import engine as E
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator().manual_seed(0)
for stage, (res, C) in enumerate(((56, 128), (28, 256), (14, 512), (7, 1024))):
    x = torch.randn((B, res, res, C), generator=g).half().to(dev)
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    win = E.layernorm_window_partition(x, gam, bet, 1e-5, 7, 3 if res > 7 else 0)
    cases = {"ln1+partition": lambda: E.layernorm_window_partition(x, gam, bet, 1e-5, 7, 3 if res > 7 else 0),
             "reverse+res+ln2": lambda: E.window_reverse_layernorm(win, x, gam, bet, 1e-5, 7, 3 if res > 7 else 0),
             "plain layernorm": lambda: E.layernorm(x, gam, bet, 1e-5)}
    mb = x.numel() * 2 / 1e6
    out = []
    for k, f in cases.items():
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()          # graph replay: the host is out of the loop (a Python call costs more than these kernels)
        with torch.cuda.graph(gr):
            for _ in range(10):
                f()
        gr.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gr.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(100 * e0.elapsed_time(e1))
        n = 4 if k.startswith("reverse") else 2
        t = sorted(ts)[2]
        out.append(f"{k}: {t:.1f} us ({n * mb / t:.2f} TB/s)")
    print(f"stage {stage + 1} ({mb:.1f} MB per tensor): " + "   ".join(out), flush=True)
