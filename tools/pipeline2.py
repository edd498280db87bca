"""Experiment: consecutive forwards (each already two half batches on two streams) issued alternately on two launch streams, so
that the ramp-down of one overlaps the ramp-up of the next.  usage: pipeline2.py [workload] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import seeded, models
from tlxcv_amd.graph import GraphedForward

wl = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
ctor = {"vit_b16": "vit_base_patch16_224", "swin_b": "swintransformer_base_patch4_window7_224"}.get(wl, wl)
m = getattr(models, ctor)()
m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
m = m.to(dev).set_eval()
x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(bs // 32, 1, 1, 1).contiguous()
g1 = GraphedForward(m, x.clone())
g2 = GraphedForward(m, x.clone())
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()


def run(n, two):
    for i in range(n):
        if two:
            with torch.cuda.stream(s1 if i % 2 == 0 else s2):
                (g1 if i % 2 == 0 else g2)()
        else:
            g1()


for two in (False, True, False, True):
    run(10, two)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        run(20, two)
        s1.synchronize(); s2.synchronize()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    print(f"{wl} batch {bs} {'two launch streams alternating' if two else 'one launch stream'}: {sorted(ts)[2]:.3f} ms / forward", flush=True)
