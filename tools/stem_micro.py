"""Interleaved A/B: the ResNet stem + max-pool as one launch vs two, batch 256 (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import _lib
_lib.tuning().__enter__()      # libtlxmi_tune.so: the flavour that reads the TLXMI_* knobs
from tlxcv_amd import models, seeded

dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
m = models.resnet50()
m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
m = m.to(dev).set_eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(B // 32, 1, 1, 1).contiguous()


def run(fused):
    return m.conv1.run_stem(x, 2, m.bn1, 1, maxpool=m.maxpool) if fused else m.maxpool.run_nhwc(m.conv1.run_stem(x, 2, m.bn1, 1))


for f in (0, 1):
    run(f)
torch.cuda.synchronize()
res = {0: [], 1: []}
for r in range(10):
    for f in (0, 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run(f)
        e1.record()
        torch.cuda.synchronize()
        res[f].append(e0.elapsed_time(e1) / 5 * 1e3)
for f in (0, 1):
    v = sorted(res[f])
    print(f"{'fused' if f else 'two launches'}: median {v[len(v)//2]:.1f} us  min {v[0]:.1f} us  (layout kernel included)")
assert os.environ.get('TLXMI_DEBUG', '0') not in ('0', '4', '8', '12') or torch.equal(run(0), run(1))
