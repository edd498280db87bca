"""What bounds the persistent 256 x 256 GEMM: three launch kinds for tools/pmc_wall.sh to tell apart by launch order —
(A) K = 3072, no epilogue arithmetic and no stores (TLXMI_DEBUG=3): the bare operand stream + MFMA loop,
(B) K = 3072 as shipped, (C) K = 768 as shipped (ViT qkv).  M = 50432, N = 2304, fp16, bias epilogue.  Tuning flavour.
Prints the wall-clock rate of each (unprofiled runs only mean something for that)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import _lib
_lib.tuning().__enter__()
from tlxcv_amd import engine as E

dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
M, N = 50432, 2304
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
os.environ["TLXMI_TILE"] = "8"
for tag, K, dbg in (("A bare K=3072", 3072, "3"), ("B shipped K=3072", 3072, "0"), ("C shipped K=768", 768, "0")):
    g = torch.Generator().manual_seed(1)
    x = torch.randn((M, K), generator=g).half().to(dev)
    pk = E.PackedFilter((torch.randn((N, K), generator=g) * K ** -0.5).view(N, K, 1, 1).to(dev), torch.float16)
    b = torch.randn(N, generator=g).to(dev)
    os.environ["TLXMI_DEBUG"] = dbg
    E.linear(x, pk, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        E.linear(x, pk, b)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / REPS * 1e3
    print(f"{tag}: {us:8.1f} us  {2.0 * M * K * N / us / 1e6:6.0f} TFLOP/s", flush=True)
