"""Print the dispatcher's tile choice per layer shape for one forward (tuning flavour, TLXMI_TRACE_TILES).  usage: trace_tiles_wl.py ctor batch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tlxcv_amd
from tlxcv_amd import seeded, models, _lib
_lib.tuning().__enter__()
wl = sys.argv[1] if len(sys.argv) > 1 else "swin_b"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
ctor = {"vit_b16": "vit_base_patch16_224", "swin_b": "swintransformer_base_patch4_window7_224"}.get(wl, wl)
m = getattr(models, ctor)()
m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
m = m.to(dev).set_eval()
x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(bs // 32, 1, 1, 1).contiguous()
os.environ["TLXMI_TRACE_TILES"] = "0"
m(x); torch.cuda.synchronize()
os.environ["TLXMI_TRACE_TILES"] = sys.argv[3] if len(sys.argv) > 3 else "1"
m(x); torch.cuda.synchronize()
