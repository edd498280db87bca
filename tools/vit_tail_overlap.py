"""Does the LayerNorm that follows proj / fc2 hide under the GEMM's tail launch?  ViT-B/16 batch 256 (M = 50432):
A: linear(+res) over all rows (the dispatcher's main + tail launches), then LayerNorm over all rows.
B: rows split on the host at the dispatcher's own boundary: main GEMM | tail GEMM on the caller's stream with the LayerNorm of the
   main rows on a side stream beside it | LayerNorm of the tail rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlxcv_amd import engine as E
dev = torch.device("cuda:0")
M, D = 256 * 197, 768
g = torch.Generator().manual_seed(0)
gamma, beta = (torch.rand(D, generator=g) + 0.5).to(dev), (torch.randn(D, generator=g) * 0.1).to(dev)
side = torch.cuda.Stream()
MS = 170 * 256          # rows of the whole rounds: 170 M tiles x 3 N tiles = 510 tiles <= 2 x 256
for name, K in (("proj", 768), ("fc2", 3072)):
    x = torch.randn((M, K), generator=g).half().to(dev)
    res = torch.randn((M, D), generator=g).half().to(dev)
    w = (torch.randn((D, K), generator=g) * K ** -0.5).to(dev)
    b = (torch.randn(D, generator=g) * 0.1).to(dev)
    pk = E.PackedFilter(w, torch.float16)
    out = torch.empty_like(res)
    yn = torch.empty_like(res)

    def A():
        y = E.linear(x, pk, b, res=res, out=out)
        return E.layernorm(y, gamma, beta, 1e-6)

    def B():
        cur = torch.cuda.current_stream()
        E.linear(x[:MS], pk, b, res=res[:MS], out=out[:MS])
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            n0 = E.layernorm(out[:MS], gamma, beta, 1e-6)
        E.linear(x[MS:], pk, b, res=res[MS:], out=out[MS:])
        n1 = E.layernorm(out[MS:], gamma, beta, 1e-6)
        cur.wait_stream(side)
        return n0, n1

    def Bserial():
        E.linear(x[:MS], pk, b, res=res[:MS], out=out[:MS])
        n0 = E.layernorm(out[:MS], gamma, beta, 1e-6)
        E.linear(x[MS:], pk, b, res=res[MS:], out=out[MS:])
        n1 = E.layernorm(out[MS:], gamma, beta, 1e-6)
        return n0, n1

    ya = A()
    n0, n1 = B()
    torch.cuda.synchronize()
    assert torch.equal(ya[:MS], n0) and torch.equal(ya[MS:], n1), "split changes the result"
    graphs = {}
    for k, f in (("A one piece", A), ("B tail beside LayerNorm", B), ("B serial", Bserial)):
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            f()
        graphs[k] = gr
    ts = {k: [] for k in graphs}
    for rep in range(9):
        for k, gr in graphs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
            ts[k].append(1e3 * e0.elapsed_time(e1) / 5)
    print(name, "  ".join(f"{k}: {sorted(v)[len(v) // 2]:.1f} us" for k, v in ts.items()), flush=True)
