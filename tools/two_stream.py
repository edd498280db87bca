"""Experiment: one batch-256 forward vs two half batches on two HIP streams (inside one hipGraph): does an MFMA-bound kernel of
one half overlap an HBM-bound kernel of the other?   usage: two_stream.py [workload] [batch] [parts]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import seeded, models

wl = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
ctor = {"vit_b16": "vit_base_patch16_224", "swin_b": "swintransformer_base_patch4_window7_224"}.get(wl, wl)
m = getattr(models, ctor)()
m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
m = m.to(dev).set_eval()
x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(bs // 32, 1, 1, 1).contiguous()


SIDE = [torch.cuda.Stream() for _ in range(3)]


def run_parts(parts, skew=None):
    if parts == 1:
        return m(x)
    cur = torch.cuda.current_stream()
    n = bs // parts
    outs = [None] * parts
    side = SIDE[:parts - 1]
    for i, s in enumerate(side):
        s.wait_stream(cur)
    if skew is not None and parts == 2 and hasattr(m, "forward_one"):
        ev = torch.cuda.Event()

        def mark(i):
            if i == skew:
                ev.record(cur)
        outs[0] = m.forward_one(x[:n], mark)
        side[0].wait_event(ev)
    else:
        outs[0] = m(x[:n])
    for i, s in enumerate(side):
        with torch.cuda.stream(s):
            outs[i + 1] = m(x[(i + 1) * n:(i + 2) * n])
    for s in side:
        cur.wait_stream(s)
    return torch.cat(outs, 0)


ref = run_parts(1).float()
keep = []
for parts, skew in ((1, None), (2, None), (1, None), (2, None)):
    for _ in range(3):
        y = run_parts(parts, skew)
    torch.cuda.synchronize()
    err = float((y.float() - ref).abs().max())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = run_parts(parts, skew)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ts.sort()
    keep.append(g)
    print(f"{wl} batch {bs} in {parts} part(s), skew {skew}: {ts[len(ts) // 2]:.3f} ms / forward  ({bs / ts[len(ts) // 2] * 1e3:.0f} img/s)  max |diff| vs 1 part {err:.3g}", flush=True)
torch.cuda.synchronize()
del keep, g
torch.cuda.synchronize()
