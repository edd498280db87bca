"""ResNet-50 batch 256 (hipGraph replay, interleaved): the whole forward in two halves (shipped) vs the stem and the first k
bottleneck blocks in halves and the rest (the 7 x 7 stage: 25 row tiles per half) on the whole batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import seeded, models, engine as E
from tlxcv_amd.models.classification.resnet import run_bottleneck_chain
dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = models.resnet50()
m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
m = m.to(dev).set_eval()
x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(bs // 32, 1, 1, 1).contiguous()
blocks = [blk for layer in (m.layer1, m.layer2, m.layer3, m.layer4) for blk in layer]


def head(h, k):
    v = m.conv1.run_stem(h, 2, m.bn1, E.ACT_RELU, maxpool=m.maxpool)
    return run_bottleneck_chain(blocks[:k], v)


def hybrid(k, tail_plan):
    def f():
        E.set_option("two_streams", False)
        v = E.run_halves(lambda h: head(h, k), x, "half")
        if tail_plan is None:
            v = run_bottleneck_chain(blocks[k:], v)
        else:
            with E.shared_plan(tail_plan):
                v = run_bottleneck_chain(blocks[k:], v)
        return m.fc.run(E.global_avgpool(v))
    return f


def shipped():
    E.set_option("two_streams", True)
    return m(x)


cfgs = {"shipped": shipped}
for k in (13, 7):
    cfgs[f"{k} blocks in halves"] = hybrid(k, None)
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
    print(f"{name:28s} {sorted(t)[3]:.3f} ms   max |logit - shipped| {graphs[name][1]:.4f}", flush=True)
