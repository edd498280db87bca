import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, tlxcv_amd
from tlxcv_amd import engine as E
dev = torch.device("cuda:0")
for prec, dt in (("fp16", torch.float16), ("fp32", torch.float32)):
    tlxcv_amd.set_precision(prec)
    for (N, H, C, Co, R, s, p) in ((128, 7, 512, 512, 3, 1, 1), (32, 7, 512, 512, 3, 1, 1), (8, 14, 256, 256, 3, 1, 1), (128, 14, 512, 512, 3, 2, 1), (64, 14, 1024, 2048, 1, 2, 0), (4, 7, 512, 512, 3, 1, 1)):
        g = torch.Generator().manual_seed(5)
        x = torch.randn((N, H, H, C), generator=g).to(dt).to(dev)
        w = (torch.randn((Co, C, R, R), generator=g) * (2 / (C * R * R)) ** 0.5)
        pk = E.PackedFilter(w.to(dev), dt)
        sc, sh = torch.rand(Co, generator=g).to(dev) + 0.5, torch.randn(Co, generator=g).to(dev)
        Ho = (H + 2 * p - R) // s + 1
        res = torch.randn((N, Ho, Ho, Co), generator=g).to(dt).to(dev)
        outs = {}
        ts = {}
        for on in (1, 0):
            E.set_option("conv_splitk", on)
            y = E.conv2d(x, pk, s, p, 1, sc, sh, res, E.ACT_RELU)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): E.conv2d(x, pk, s, p, 1, sc, sh, res, E.ACT_RELU)
            e1.record(); torch.cuda.synchronize()
            outs[on], ts[on] = y.float(), e0.elapsed_time(e1) / 20 * 1e3
        ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.to(dev).to(dt).float(), None, s, p).permute(0, 2, 3, 1)
        ref = torch.relu(ref * sc + sh + res.float())
        E.set_option("conv_splitk", 1)
        d = E._lib.ConvDesc  # noqa
        print(f"{prec} N{N} {H}x{H} {C}->{Co} k{R} s{s}: split {ts[1]:6.1f} us  one launch {ts[0]:6.1f} us   |split - ref| {float((outs[1]-ref).abs().max()):.2e}  |one - ref| {float((outs[0]-ref).abs().max()):.2e}", flush=True)
