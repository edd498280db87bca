import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tlxcv_amd
from tlxcv_amd import _lib, engine as E
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for B in (256, 128):
    qkv = torch.randn((B, 197, 2304), generator=g).half().to(dev)
    res = {}
    for rep in range(5):
        for dbg in ("0", "1", "2", "3"):
            with _lib.tuning(TLXMI_ATTN_DBG=dbg):
                E.attention(qkv, 12, 0.125)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): E.attention(qkv, 12, 0.125)
                e1.record(); torch.cuda.synchronize()
                res.setdefault(dbg, []).append(e0.elapsed_time(e1) * 100)
    print(f"attention B={B}: " + "  ".join(f"dbg{d} {sorted(v)[2]:6.1f} us" for d, v in res.items()), flush=True)
