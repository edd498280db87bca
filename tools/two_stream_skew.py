"""Experiment: two half batches on two streams, the second starting once the first has enqueued `k` library calls (skew)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tlxcv_amd
from tlxcv_amd import seeded, models, _lib, engine as E

wl = sys.argv[1] if len(sys.argv) > 1 else "vit_b16"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
skews = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1, 2, 3, 4, 5, 6, 8]
dev = torch.device("cuda:0")
tlxcv_amd.set_precision("fp16")
E.set_option("two_streams", False)
ctor = {"vit_b16": "vit_base_patch16_224", "swin_b": "swintransformer_base_patch4_window7_224"}.get(wl, wl)
m = getattr(models, ctor)()
m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
m = m.to(dev).set_eval()
x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(bs // 32, 1, 1, 1).contiguous()
SIDE = torch.cuda.Stream()
PLAN = sys.argv[4] if len(sys.argv) > 4 else {"swin_b": "full"}.get(wl, "half")
orig_call = _lib.call
state = {"n": 0, "k": -1, "ev": None, "cur": None}


def counting_call(name, *args):
    r = orig_call(name, *args)
    if state["k"] >= 0:
        state["n"] += 1
        if state["n"] == state["k"]:
            state["ev"].record(state["cur"])
    return r


_lib.call = counting_call


def run(skew):
    if skew is None:
        return m(x)
    cur = torch.cuda.current_stream()
    n = bs // 2
    SIDE.wait_stream(cur)
    ev = torch.cuda.Event()
    state.update(n=0, k=skew if skew > 0 else -1, ev=ev, cur=cur)
    with E.shared_plan(PLAN):      # the planning hint run_halves() gives the shipped forward
        y0 = m(x[:n])
        state["k"] = -1
        if skew > 0:
            SIDE.wait_event(ev)
        with torch.cuda.stream(SIDE):
            y1 = m(x[n:])
    cur.wait_stream(SIDE)
    return torch.cat((y0, y1), 0)


keep = []
for skew in [None] + skews:
    for _ in range(3):
        y = run(skew)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = run(skew)
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
    print(f"{wl} batch {bs} skew {skew}: {ts[len(ts) // 2]:.3f} ms / forward ({bs / ts[len(ts) // 2] * 1e3:.0f} img/s)", flush=True)
torch.cuda.synchronize()
