"""Ablation of the small-block grouped-conv kernel (tuning flavour, TLXMI_GCONV_DBG bits: 1 no refill of the LDS tiles, 2 no stores,
4 no LDS reads / MFMAs) and the workgroups-per-CU knob, ResNeXt-50 stage-1 / stage-2 layers at batch 256."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlxcv_amd import engine as E, _lib
from gconv_micro import timeit
dev = torch.device("cuda:0")
for C, groups, stride, hw in ((128, 32, 1, 56), (256, 32, 2, 56), (256, 32, 1, 28), (512, 32, 2, 28)):
    cg = C // groups
    x = torch.randn((256, hw, hw, C), device=dev).half()
    w = torch.randn((C, cg, 3, 3), device=dev) * (2.0 / (cg * 9)) ** 0.5
    pk = E.PackedGroupFilter(w, groups, torch.float16)
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    f = lambda: E.group_conv2d(x, pk, stride, 1, 1, sc, sh, None, E.ACT_RELU)      # noqa: E731
    out = [f"{C} ch cg {cg} s{stride} {hw}:"]
    for dbg in (0, 1, 2, 4, 3, 5, 6, 7):
        out.append(f"dbg{dbg} {timeit(f, dict(TLXMI_GCONV_DBG=dbg)):.0f}")
    for wgs in (1, 2, 3, 4):
        out.append(f"wgs{wgs} {timeit(f, dict(TLXMI_GCONV_WGS=wgs)):.0f}")
    print("  ".join(out), flush=True)
