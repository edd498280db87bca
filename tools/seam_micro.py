"""Interleaved A/B per ResNet-50 stage, batch 256: the bottleneck seam as one launch (per TLXMI_SEAM variant) vs conv3 + skip and
conv1 as two.  usage: seam_micro.py [batch] [variants, comma separated: 0,8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import _lib
PRODUCT = os.environ.get("SEAM_PRODUCT") == "1"       # the product library (no knobs): variants collapse to "0"
if not PRODUCT:
    _lib.tuning().__enter__()
from tlxcv_amd import engine as E

dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
VARS = ["0"] if PRODUCT else [v for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0"])]
for K1, N1, N2, hw in ((64, 256, 64, 56), (64, 256, 128, 56), (128, 512, 128, 28), (128, 512, 256, 28), (256, 1024, 256, 14)):
    g = torch.Generator().manual_seed(1)
    t2 = torch.randn((B, hw, hw, K1), generator=g).half().to(dev)
    skip = torch.randn((B, hw, hw, N1), generator=g).half().to(dev)
    pk3 = E.PackedFilter((torch.randn((N1, K1, 1, 1), generator=g) * (2 / K1) ** 0.5).to(dev), torch.float16)
    pk1 = E.PackedFilter((torch.randn((N2, N1, 1, 1), generator=g) * (2 / N1) ** 0.5).to(dev), torch.float16)
    s3, h3, s1, h1 = (torch.rand(n, generator=g).to(dev) for n in (N1, N1, N2, N2))

    def run(f):
        if f != "two":
            os.environ["TLXMI_SEAM"] = f
            return E.bottleneck_seam(t2, pk3, s3, h3, skip, pk1, s1, h1)
        y = E.conv2d(t2, pk3, 1, 0, 1, s3, h3, skip, E.ACT_RELU)
        return y, E.conv2d(y, pk1, 1, 0, 1, s1, h1, None, E.ACT_RELU)
    cases = ["two"] + VARS
    outs = {f: run(f) for f in cases}
    torch.cuda.synchronize()
    same = {f: bool(torch.equal(outs[f][0], outs[VARS[0]][0]) and torch.equal(outs[f][1], outs[VARS[0]][1])) for f in VARS}
    dz = float((outs["two"][1].float() - outs[VARS[0]][1].float()).abs().max())
    del outs
    res = {f: [] for f in cases}
    for r in range(8):
        for f in cases:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run(f)
            e1.record()
            torch.cuda.synchronize()
            res[f].append(e0.elapsed_time(e1) / 5 * 1e3)
    m = B * hw * hw
    fb = m * (K1 + 2 * N1 + N2) * 2
    med = {f: sorted(v)[len(v) // 2] for f, v in res.items()}
    print(f"{K1:4d} -> {N1:4d} -> {N2:4d} @ {hw}x{hw} ({fb / 1e6:.0f} MB): two launches {med['two']:7.1f} us  " +
          "  ".join(f"seam[{f}] {med[f]:7.1f} us ({fb / med[f] / 1e6:.2f} TB/s{'' if same[f] else ' DIFFERS'})" for f in VARS) +
          f"  |t1 - two-launch| max {dz:.3g}", flush=True)
