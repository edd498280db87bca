"""gemm_w4 (candidate 11, four waves with 128 x 128 wave tiles) against the product's own choice on the ViT-B/16 / Swin-B Linear
shapes: correctness against a torch fp32 matmul of the same fp16 inputs, then interleaved timing (HIP events, median)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import _lib, engine as E

dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
g = torch.Generator().manual_seed(0)
SHAPES = [("qkv", 50432, 768, 2304, E.ACT_NONE, False), ("proj", 50432, 768, 768, E.ACT_NONE, True),
          ("fc1", 50432, 768, 3072, E.ACT_GELU, False), ("fc2", 50432, 3072, 768, E.ACT_NONE, True),
          ("long K", 50432, 3072, 2304, E.ACT_NONE, False),
          ("swin3 qkv", 25088, 512, 1536, E.ACT_NONE, False), ("swin3 fc1", 25088, 512, 2048, E.ACT_GELU, False),
          ("swin3 fc2", 25088, 2048, 512, E.ACT_NONE, True), ("swin3 proj", 25088, 512, 512, E.ACT_NONE, True)]
only = sys.argv[1].split(",") if len(sys.argv) > 1 else None
cands = os.environ.get("CANDS", "11").split(",")
for name, M, K, N, act, with_res in SHAPES:
    if only and not any(o in name for o in only):
        continue
    x = torch.randn((M, K), generator=g).half().to(dev)
    w = (torch.randn((N, K), generator=g) * K ** -0.5).to(dev)
    b = (torch.randn(N, generator=g) * 0.1).to(dev)
    res = torch.randn((M, N), generator=g).half().to(dev) if with_res else None
    pk = E.PackedFilter(w, torch.float16)
    run = lambda: E.linear(x, pk, b, act=act, res=res)
    ref = x[:4096].float() @ w.half().float().t() + b
    if with_res:
        ref = ref + res[:4096].float()
    if act == E.ACT_GELU:
        ref = torch.nn.functional.gelu(ref)
    variants = {"product": (lambda f: f())}
    for c in cands:
        def mk(c):
            def call(f):
                with _lib.tuning(TLXMI_TILE=c):
                    return f()
            return call
        variants["cand" + c] = mk(c)
    outs = {}
    for vn, call in variants.items():
        y = call(run)
        torch.cuda.synchronize()
        outs[vn] = y
        err = (y[:4096].float() - ref).abs().max().item()
        tail = (y[-300:].float() - ((x[-300:].float() @ w.half().float().t() + b) + (res[-300:].float() if with_res else 0)
                                    if act != E.ACT_GELU else torch.nn.functional.gelu(x[-300:].float() @ w.half().float().t() + b))).abs().max().item()
        print(f"{name:10s} {vn:8s} max|err| head {err:.2e} tail {tail:.2e}", flush=True)
    times = {vn: [] for vn in variants}
    for _ in range(7):
        for vn, call in variants.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                call(run)
            e1.record()
            torch.cuda.synchronize()
            times[vn].append(1e3 * e0.elapsed_time(e1) / 5)
    fl = 2.0 * M * K * N
    print(f"{name:10s} M={M} K={K} N={N}: " + "   ".join(f"{vn} {sorted(v)[3]:7.1f} us {fl / sorted(v)[3] / 1e6:5.0f} TF/s" for vn, v in times.items()), flush=True)
