"""ViT-B/16 at 384 x 384 (577 tokens, 12 heads x 64): attention kernel time, batch 64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlxcv_amd import engine as E
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
qkv = torch.randn((B, 577, 2304), device=dev).half()
for _ in range(3):
    E.attention(qkv, 12, 0.125)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    E.attention(qkv, 12, 0.125)
e1.record()
torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / 20
fl = 4 * 577 * 577 * 64 * 12 * B
print(f"attention 577 tokens batch {B}: {us:.1f} us  {fl / us / 1e6:.0f} TF/s  ({qkv.numel() * 2 * 4 / 3 / us / 1e3:.0f} GB/s of q,k,v,out)")
