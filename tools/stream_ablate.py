"""Per-tile overhead of the persistent GEMM: qkv-like shapes at several K, with TLXMI_DEBUG ablation bits (1 epilogue without
arithmetic, 2 stores suppressed).  Tuning flavour."""
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
for K in (512, 768):
    g = torch.Generator().manual_seed(1)
    x = torch.randn((M, K), generator=g).half().to(dev)
    pk = E.PackedFilter((torch.randn((N, K), generator=g) * K ** -0.5).view(N, K, 1, 1).to(dev), torch.float16)
    b = torch.randn(N, generator=g).to(dev)
    res = {}
    for tile in ("8", "7"):
        for dbg in ("0", "1", "2", "3", "4"):
            os.environ["TLXMI_TILE"] = tile
            os.environ["TLXMI_DEBUG"] = dbg
            E.linear(x, pk, b)
            ts = []
            for rep in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    E.linear(x, pk, b)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 5 * 1e3)
            res[(tile, dbg)] = sorted(ts)[2]
    fl = 2.0 * M * K * N
    print(f"K {K:5d} ({K // 64:3d} K tiles): " + "  ".join(f"cand{t}/dbg{d} {v:7.1f} us {fl / v / 1e6:5.0f} TF/s" for (t, d), v in res.items()), flush=True)
os.environ.pop("TLXMI_TILE"); os.environ.pop("TLXMI_DEBUG")
