"""ViT-B / Swin-B Linear layers one by one (batch 256 / 128 tokens): us and TFLOP/s.  FC_PRODUCT=1: the product library
(compare two builds by swapping libtlxmi.so); otherwise the tuning flavour, sweeping TLXMI_GELU_STREAM for the GELU layers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import _lib
PRODUCT = os.environ.get("FC_PRODUCT") == "1"
if not PRODUCT:
    _lib.tuning().__enter__()
from tlxcv_amd import engine as E

dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
CASES = [("vit qkv", 50432, 768, 2304, E.ACT_NONE, False), ("vit proj", 50432, 768, 768, E.ACT_NONE, True),
         ("vit fc1", 50432, 768, 3072, E.ACT_GELU, False), ("vit fc2", 50432, 3072, 768, E.ACT_NONE, True),
         ("swin1 fc1", 401408, 128, 512, E.ACT_GELU, False), ("swin2 fc1", 100352, 256, 1024, E.ACT_GELU, False),
         ("swin3 fc1", 25088, 512, 2048, E.ACT_GELU, False), ("swin4 fc1", 6272, 1024, 4096, E.ACT_GELU, False)]
for name, M, K, N, act, res in CASES:
    g = torch.Generator().manual_seed(1)
    x = torch.randn((M, K), generator=g).half().to(dev)
    w = (torch.randn((N, K), generator=g) * K ** -0.5)
    pk = E.PackedFilter(w.view(N, K, 1, 1).to(dev), torch.float16)
    b = torch.randn(N, generator=g).to(dev)
    r = torch.randn((M, N), generator=g).half().to(dev) if res else None
    vars_ = ["0"] if PRODUCT or act != E.ACT_GELU else ["0", "1"]
    out = {}
    for v in vars_:
        os.environ["TLXMI_GELU_STREAM"] = v
        out[v] = E.linear(x, pk, b, r, act)
    torch.cuda.synchronize()
    ref = x[:512].float() @ w.to(dev).half().float().t() + b
    if act == E.ACT_GELU:
        ref = torch.nn.functional.gelu(ref)
    if res:
        ref = ref + r[:512].float()
    err = {v: float((o[:512].float() - ref).abs().max()) for v, o in out.items()}
    ts = {v: [] for v in vars_}
    for rep in range(6):
        for v in vars_:
            os.environ["TLXMI_GELU_STREAM"] = v
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                E.linear(x, pk, b, r, act)
            e1.record()
            torch.cuda.synchronize()
            ts[v].append(e0.elapsed_time(e1) / 5 * 1e3)
    fl = 2.0 * M * K * N
    print(f"{name:10s} M {M} K {K} N {N}: " + "  ".join(f"[{v}] {sorted(t)[len(t) // 2]:7.1f} us {fl / sorted(t)[len(t) // 2] / 1e6:6.0f} TF/s err {err[v]:.2e}" for v, t in ts.items()), flush=True)
