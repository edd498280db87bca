"""YOLOv3 (DarkNet-53, 416 px, batch 32) forward under rocprofv3: where the detection post-process stands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import seeded, models
dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = models.YOLOv3()
m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
m = m.to(dev).set_eval()
x = torch.from_numpy(seeded.image_batch(16, 0, hw=416)).to(dev).repeat(bs // 16, 1, 1, 1).contiguous()
for _ in range(3):
    out = m({"images": x})
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    out = m({"images": x})
e1.record()
torch.cuda.synchronize()
print(f"YOLOv3 batch {bs}: {e0.elapsed_time(e1) / 5:.2f} ms; outputs {[ (k, tuple(v.shape)) for k, v in out.items() if hasattr(v, 'shape')]}")
