"""Is round 2's LayerNorm-folded GEMM (tlxmi_row_stats + tlxmi_linear_ln, gemm_stream ROWAFF) bit-reproducible?  Runs the r02
build of the library (tools/probe/libtlxmi_r02.so, built from commit 2890fbd) next to the current one."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlxcv_amd import engine as E
dev = torch.device("cuda:0")
old = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libtlxmi_r02.so"))
M, D = int(sys.argv[1]) if len(sys.argv) > 1 else 13199, 768
g = torch.Generator().manual_seed(0)
x = (torch.randn((M, D), generator=g) * 1.2 + 0.1).half().to(dev)
gamma, beta = (torch.rand(D, generator=g) + 0.5).to(dev), (torch.randn(D, generator=g) * 0.1).to(dev)
vp = C.c_void_p
for cout in (3072, 2304, 768):
    w = (torch.randn((cout, D), generator=g) * D ** -0.5).to(dev)
    b = (torch.randn(cout, generator=g) * 0.1).to(dev)
    prep = E.LinearLN(w, b, gamma, beta, torch.float16)
    stats = torch.empty((M, 2), dtype=torch.float32, device=dev)
    E._lib.call("tlxmi_row_stats", vp(x.data_ptr()), 0, M, D, D, C.c_float(1e-6), vp(stats.data_ptr()), None)
    torch.cuda.synchronize()

    def run(lib):
        y = torch.empty((M, cout), dtype=torch.float16, device=dev)
        rc = lib.tlxmi_linear_ln(C.c_int(0), C.c_int64(M), C.c_int(D), C.c_int(cout), C.c_int(D), C.c_int(cout), vp(x.data_ptr()), vp(prep.pk.buf.data_ptr()),
                                 vp(prep.c1.data_ptr()), vp(prep.c2.data_ptr()), vp(stats.data_ptr()), C.c_int(0), vp(y.data_ptr()), None)
        assert rc == 0, rc
        torch.cuda.synchronize()
        return y
    for name, lib in (("r02", old), ("now", E._lib.load())):
        ys = [run(lib) for _ in range(8)]
        nd = [int((y != ys[0]).sum()) for y in ys[1:]]
        print(f"M={M} Cout={cout} {name}: elements differing from run 0: {nd}", flush=True)
