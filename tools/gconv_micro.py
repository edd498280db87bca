"""ResNeXt-50 32x4d grouped 3x3 layers at batch 256 (hipGraph replay): the small-block MFMA kernel (group_conv.hip) vs the
block-diagonal implicit GEMM (TLXMI_GCONV=0, tuning flavour).  HBM floor = input + output bytes at 5 TB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlxcv_amd import engine as E, _lib
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def timeit(f, env):
    with _lib.tuning(**env):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(5):
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
        ts.append(200 * e0.elapsed_time(e1))
    return sorted(ts)[3]


if __name__ == "__main__":
  for C, groups, stride, hw in ((128, 32, 1, 56), (256, 32, 2, 56), (256, 32, 1, 28), (512, 32, 2, 28), (512, 32, 1, 14), (1024, 32, 2, 14),
                                (1024, 32, 1, 7), (256, 64, 1, 56), (512, 64, 1, 28)):
      cg = C // groups
      x = torch.randn((B, hw, hw, C), device=dev).half()
      w = torch.randn((C, cg, 3, 3), device=dev) * (2.0 / (cg * 9)) ** 0.5
      pk = E.PackedGroupFilter(w, groups, torch.float16)
      sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
      f = lambda: E.group_conv2d(x, pk, stride, 1, 1, sc, sh, None, E.ACT_RELU)      # noqa: E731
      ho = (hw - 1) // stride + 1
      byt = (B * hw * hw * C + B * ho * ho * C) * 2
      new, old = timeit(f, {}), timeit(f, dict(TLXMI_GCONV=0))
      print(f"{C:5d} ch, {groups} groups of {cg:2d}, stride {stride}, {hw:2d} x {hw:2d}: block-diagonal {old:7.1f} us   4x4x4 MFMA {new:7.1f} us   "
            f"({byt / 1e6:.0f} MB = {byt / 5e6:.1f} us at 5 TB/s)", flush=True)
