"""Swin-B batch 128 (hipGraph replay, interleaved): the whole forward in two halves on two streams (shipped) vs hybrids — the first k
stages in halves, the rest on the whole batch (stage 3's GEMMs have 49 row tiles per half: fewer tiles than CUs for proj / fc2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import seeded, models, engine as E
from tlxcv_amd.tlx import nn
dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
m = models.swintransformer_base_patch4_window7_224()
m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
m = m.to(dev).set_eval()
x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(bs // 32, 1, 1, 1).contiguous()


def tail(t, k):
    for layer in m.layers[k:]:
        t = layer.run(t)
    t = m.norm(t)
    return m.head.run(E.global_avgpool(t))


def head_part(h, k):
    t = m.patch_embed(h)
    for layer in m.layers[:k]:
        t = layer.run(t)
    return t


def hybrid(k, plan, plan_tail):
    def f():
        E.set_option("two_streams", False)
        t = E.run_halves(lambda h: head_part(h, k), x, plan)
        if plan_tail is None:
            return tail(t, k)
        with E.shared_plan(plan_tail):
            return tail(t, k)
    return f


def shipped():
    E.set_option("two_streams", True)
    return m(x)


def one():
    E.set_option("two_streams", False)
    return m(x)


cfgs = {"shipped (4 stages in halves)": shipped, "one stream": one}
for k in (1, 2, 3):
    cfgs[f"{k} stages in halves"] = hybrid(k, "full", None)
ref = shipped().float()
graphs = {}
for name, f in cfgs.items():
    for _ in range(3):
        y = f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = f()
    g.replay()
    torch.cuda.synchronize()
    graphs[name] = (g, (y.float() - ref).abs().max().item())
ts = {k: [] for k in graphs}
for rep in range(7):
    for name, (g, _) in graphs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts[name].append(e0.elapsed_time(e1) / 5)
for name, t in ts.items():
    print(f"{name:32s} {sorted(t)[3]:.3f} ms   max |logit - shipped| {graphs[name][1]:.4f}", flush=True)
