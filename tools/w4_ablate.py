"""Timing ablations of gemm_w4's K loop (tuning flavour, TLXMI_W4_DBG bits: 1 no LDS-DMA, 2 no fragment reads, 4 no mid-tile wait +
barrier, 8 no MFMAs) on the long-K shape; results are wrong by construction."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import _lib, engine as E
dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
g = torch.Generator().manual_seed(0)
for M, K, N in ((50432, 3072, 2304), (50432, 768, 2304)):
    x = torch.randn((M, K), generator=g).half().to(dev)
    pk = E.PackedFilter((torch.randn((N, K), generator=g) * K ** -0.5).to(dev), torch.float16)
    b = torch.randn(N, generator=g).to(dev)
    res = {}
    for rep in range(3):
        for dbg in ("0", "16", "32", "64", "128", "192", "320"):
            with _lib.tuning(TLXMI_TILE="11", TLXMI_W4_DBG=dbg):
                E.linear(x, pk, b)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    E.linear(x, pk, b)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(dbg, []).append(e0.elapsed_time(e1) / 4 * 1e3)
    kt = ((M + 255) // 256) * ((N + 255) // 256) * (K // 64) / 256.0
    print(f"M={M} K={K} N={N} ({kt:.0f} K tiles per CU): " + "  ".join(f"dbg{d} {sorted(v)[1]:7.1f} us ({sorted(v)[1] / kt:5.2f}/Kt)" for d, v in res.items()), flush=True)
