import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tlxcv_amd
from tlxcv_amd import seeded, engine as E
dev = torch.device("cuda:0")
for B in (128, 256):
    x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(B // 32, 1, 1, 1).contiguous()
    y = E.nchw_to_nhwc_s2d(x, 2, torch.float16)
    # reference: torch
    ref = x.reshape(B, 3, 112, 2, 112, 2).permute(0, 2, 4, 3, 5, 1).reshape(B, 112, 112, 12).half()
    print("equal", torch.equal(y[..., :12], ref), "pad zero", not y[..., 12:].any())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        E.nchw_to_nhwc_s2d(x, 2, torch.float16)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"B={B}: {us:.1f} us  {(x.numel()*4 + y.numel()*2)/us/1e6:.2f} TB/s")
