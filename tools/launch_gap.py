"""Cost of a dependent launch inside a hipGraph on this box: a chain of n tiny kernels on one stream, and two such chains on two
streams (the forwards' shape).  usage: launch_gap.py"""
import torch
dev = torch.device("cuda:0")
a = torch.zeros(64, device=dev)
b = torch.zeros(64, device=dev)
side = torch.cuda.Stream(device=dev)


def chain(n, two):
    cur = torch.cuda.current_stream(dev)
    if two:
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(n):
                b.add_(1.0)
    for _ in range(n):
        a.add_(1.0)
    if two:
        cur.wait_stream(side)


for n in (100, 400):
    for two in (False, True):
        chain(n, two)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            chain(n, two)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{n} tiny kernels per stream, {'two streams' if two else 'one stream'}: {ms * 1e3:.0f} us per replay = {ms * 1e3 / n:.2f} us per dependent launch")


# the same two chains as TWO linear graphs replayed on two streams (instead of one graph with two branches)
s0, s1 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
for n in (100, 400):
    gs = []
    for t, s in ((a, s0), (b, s1)):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            for _ in range(n):
                t.add_(1.0)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                for _ in range(n):
                    t.add_(1.0)
        gs.append(g)
    torch.cuda.synchronize()

    def replay():
        cur = torch.cuda.current_stream(dev)
        s0.wait_stream(cur)
        s1.wait_stream(cur)
        with torch.cuda.stream(s0):
            gs[0].replay()
        with torch.cuda.stream(s1):
            gs[1].replay()
        cur.wait_stream(s0)
        cur.wait_stream(s1)
    replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{n} tiny kernels per stream, two linear graphs on two streams: {ms * 1e3:.0f} us per replay = {ms * 1e3 / n:.2f} us per dependent launch")
