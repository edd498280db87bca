"""Where the two-stream forward starts to pay: hipGraph replay of one forward on ONE stream against the same batch FORCED through
engine.run_halves (plans None / "half" / "full"), below and above the model's own threshold.
usage: two_stream_threshold.py [resnet50|vit_b16|swin_b] [batches=32,64,96,128]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import seeded, models, engine as E

wl = sys.argv[1] if len(sys.argv) > 1 else "swin_b"
batches = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "32,64,96,128").split(",")]
dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
ctor = {"vit_b16": "vit_base_patch16_224", "swin_b": "swintransformer_base_patch4_window7_224"}.get(wl, wl)
m = getattr(models, ctor)()
m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
m = m.to(dev).set_eval()
plain = type(m).forward.__wrapped__            # the forward under the two_streams decorator

for bs in batches:
    x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat((bs + 31) // 32, 1, 1, 1)[:bs].contiguous()
    arms = {"one stream": lambda: plain(m, x)}
    for plan in (None, "half", "full"):
        arms[f"two streams, plan {plan}"] = (lambda p: (lambda: E.run_halves(lambda h: plain(m, h), x, p)))(plan)
    graphs = {}
    for k, fn in arms.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = fn()
        graphs[k] = g
        g.replay()
    torch.cuda.synchronize()
    ts = {k: [] for k in graphs}
    for rep in range(7):
        for k, g in graphs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            ts[k].append(e0.elapsed_time(e1) / 10)
    base = sorted(ts["one stream"])[3]
    print(f"{wl} batch {bs}: " + "   ".join(f"{k} {sorted(t)[3]:.3f} ms ({100 * (sorted(t)[3] - base) / base:+.1f} %)" for k, t in ts.items()), flush=True)
