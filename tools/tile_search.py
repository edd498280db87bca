"""Coordinate descent over the dispatcher's tile candidate per layer shape, scored by the hipGraph replay of the whole
(two-stream) forward — tuning flavour, TLXMI_FORCE.  Prints the layers where another candidate beats the cost model's choice.
usage: tile_search.py [workload=resnet50] [batch=256] [passes=1]"""
import os, re, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import seeded, models, engine as E, _lib
_lib.tuning().__enter__()

wl = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
ctor = {"vit_b16": "vit_base_patch16_224", "swin_b": "swintransformer_base_patch4_window7_224"}.get(wl, wl)
m = getattr(models, ctor)()
m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
m = m.to(dev).set_eval()
x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(bs // 32, 1, 1, 1).contiguous()
for _ in range(3):
    m(x)
torch.cuda.synchronize()

# the layer shapes and the model's own choices: one traced forward (stderr of this process, via a pipe)
r, w = os.pipe()
saved = os.dup(2)
os.dup2(w, 2)
os.environ["TLXMI_TRACE_TILES"] = "1"
m(x)
torch.cuda.synchronize()
os.environ["TLXMI_TRACE_TILES"] = "0"
os.dup2(saved, 2)
os.close(w)
trace = os.fdopen(r).read()
base = {}
for mm in re.finditer(r"tile M=(\d+) K=(\d+) N=(\d+) R=(\d) s=(\d) res=\d plan_cus=\d+ -> cand (\d+)", trace):
    base.setdefault(":".join(mm.groups()[:5]), int(mm.group(6)))
print(f"{len(base)} layer shapes", flush=True)


def score(force):
    os.environ["TLXMI_FORCE"] = ",".join(f"{k}={v}" for k, v in force.items())
    try:
        for _ in range(2):
            m(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            m(x)
        g.replay()
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 8)
        del g
        return sorted(ts)[len(ts) // 2]
    except RuntimeError as e:
        return float("inf")


force = {}
best = score(force)
print(f"baseline {best:.3f} ms", flush=True)
for p in range(passes):
    for key, b in base.items():
        cur = force.get(key, b)
        for c in (0, 1, 2, 3, 4, 6, 7, 8, 9, 10):
            if c == cur:
                continue
            trial = dict(force)
            trial[key] = c
            t = score(trial)
            if t < best * 0.996:
                t2 = score(trial)          # confirm
                b2 = score(force)
                if t2 < b2 * 0.997:
                    print(f"  {key}: cand {cur} -> {c}: {b2:.3f} -> {t2:.3f} ms", flush=True)
                    force, best, cur = trial, t2, c
    print(f"pass {p}: {best:.3f} ms with {force}", flush=True)
