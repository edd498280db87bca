"""LayerNorm launch-shape sweep (tuning flavour): workgroups per CU of the persistent grid x the threshold in trips."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlxcv_amd import engine as E, _lib
dev = torch.device("cuda:0")


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


shapes = [(200704, 128), (50176, 256), (12544, 512), (3136, 1024), (50432, 768), (25088, 512), (100352, 256)]
print("product:", "  ".join(f"{r}x{c}: {timeit(lambda: E.layernorm(x, g, b, 1e-5)):.1f}" for (r, c) in shapes
                            for x, g, b in [(torch.randn((r, c), device=dev).half(), torch.ones(c, device=dev), torch.zeros(c, device=dev))]), flush=True)
for percu in (2, 4, 8):
    for trips in (1, 2, 4):
        with _lib.tuning(TLXMI_LN_PERCU=percu, TLXMI_LN_TRIPS=trips):
            out = []
            for (r, c) in shapes:
                x, g, b = torch.randn((r, c), device=dev).half(), torch.ones(c, device=dev), torch.zeros(c, device=dev)
                out.append(f"{r}x{c}: {timeit(lambda: E.layernorm(x, g, b, 1e-5)):.1f}")
            print(f"per_cu {percu} trips {trips}:", "  ".join(out), flush=True)
