"""engine.two_streams / run_halves: a large batch as two half batches on two HIP streams — every row depends on its
own image only (fp32: bit-identical to the halves run alone; fp16: to rounding, the halves alone take other launch shapes); below
the threshold, with the option off, on host tensors or with extra arguments nothing is split."""
import numpy as np
import pytest
import torch

import tlxcv_amd
from tlxcv_amd import engine as E, models, seeded


class _Probe:
    calls = []

    @E.two_streams(4)
    def forward(self, x, *args):
        _Probe.calls.append(tuple(x.shape))
        return x.float().mean(dim=(1, 2, 3), keepdim=False).view(-1, 1) * 2


def test_decorator_host_logic_passes_everything_through_that_is_not_a_large_cuda_batch():
    p = _Probe()
    _Probe.calls.clear()
    x = torch.arange(8 * 3 * 2 * 2, dtype=torch.float32).view(8, 3, 2, 2)
    y = p.forward(x)                                   # host tensor: no split (and no CUDA call)
    assert _Probe.calls == [(8, 3, 2, 2)] and y.shape == (8, 1)
    assert p.forward.__name__ == "forward"




@pytest.mark.gpu
def test_split_rules_on_the_device():
    dev = torch.device("cuda:0")
    p = _Probe()
    x = torch.randn((8, 3, 4, 4), device=dev)
    _Probe.calls.clear()
    y = p.forward(x)
    assert _Probe.calls == [(4, 3, 4, 4), (4, 3, 4, 4)]                 # two halves
    assert torch.equal(y, x.float().mean(dim=(1, 2, 3)).view(-1, 1) * 2)
    for xx, why in ((x[:2], "below the threshold"), (x[:7], "odd batch")):
        _Probe.calls.clear()
        p.forward(xx)
        assert _Probe.calls == [tuple(xx.shape)], why
    _Probe.calls.clear()
    p.forward(x, 1)                                                     # extra arguments: the plain call
    assert _Probe.calls == [(8, 3, 4, 4)]
    E.set_option("two_streams", False)
    try:
        _Probe.calls.clear()
        p.forward(x)
        assert _Probe.calls == [(8, 3, 4, 4)]
    finally:
        E.set_option("two_streams", True)


class _GraphOnly:
    calls = []

    @E.two_streams(4, eager=False)
    def forward(self, x):
        _GraphOnly.calls.append(tuple(x.shape))
        return x * 2


@pytest.mark.gpu
def test_graph_only_models_split_only_while_a_graph_is_captured():
    """eager=False (MobileNetV3, EfficientNet: host-bound kernel by kernel): one pass when launched eagerly, two halves inside
    a hipGraph capture."""
    dev = torch.device("cuda:0")
    p = _GraphOnly()
    x = torch.randn((8, 3, 4, 4), device=dev)
    _GraphOnly.calls.clear()
    y = p.forward(x)
    assert _GraphOnly.calls == [(8, 3, 4, 4)] and torch.equal(y, x * 2)
    p.forward(x)
    torch.cuda.synchronize()
    _GraphOnly.calls.clear()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = p.forward(x)
    assert _GraphOnly.calls == [(4, 3, 4, 4), (4, 3, 4, 4)]
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, x * 2)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_resnet50_batch_128_in_two_halves_equals_the_halves_alone(prec):
    dev = torch.device("cuda:0")
    tlxcv_amd.set_precision(prec)
    try:
        m = models.resnet50()
        m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
        m = m.to(dev).set_eval()
        x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(4, 1, 1, 1).contiguous()
        x[64:] = x[64:].flip(3)                                          # the halves differ
        y = m(x)                                                         # 128 images: split
        E.set_option("two_streams", False)
        try:
            y0, y1 = m(x[:64]), m(x[64:])
            whole = m(x)                                                 # one launch sequence over all 128
        finally:
            E.set_option("two_streams", True)
        torch.cuda.synchronize()
        if prec == "fp32":
            assert torch.equal(y[:64], y0) and torch.equal(y[64:], y1)
        else:
            # fp16: a half alone (one stream) may take other launch shapes than inside the two-stream forward (tile plans for half the
            # CUs; until round 5 also the 14 x 14 seams) — the same values to fp16 rounding, and every row depends on its own image only
            alone = torch.cat((y0, y1), 0).float()
            sc = max(1.0, float(alone.abs().max()))
            assert float((y.float() - alone).abs().max()) <= 3e-3 * sc
            y_swapped = m(torch.cat((x[64:], x[:64]), 0))                    # the halves change streams: bit-identical rows
            assert torch.equal(y_swapped[:64], y[64:]) and torch.equal(y_swapped[64:], y[:64])
        # against the unsplit forward: the dispatcher may pick other tiles for 128 rows than for 64 — same values to rounding
        tol = 1e-4 if prec == "fp32" else 3e-3
        scale = max(1.0, float(whole.float().abs().max()))
        assert float((y.float() - whole.float()).abs().max()) <= tol * scale
        assert torch.equal(y.float().argmax(1), whole.float().argmax(1)) or prec == "fp16"
    finally:
        tlxcv_amd.set_precision("fp32")


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_vit_batch_64_in_two_halves_equals_the_halves_alone(prec):
    """vision_transformer.py:313-334 under the two-stream decorator (round 3): every token row depends on its own image only."""
    dev = torch.device("cuda:0")
    tlxcv_amd.set_precision(prec)
    try:
        m = models.vit_small_patch16_224()
        m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
        m = m.to(dev).set_eval()
        x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(2, 1, 1, 1).contiguous()
        x[32:] = x[32:].flip(3)
        y = m(x)                                                         # 64 images: split
        E.set_option("two_streams", False)
        try:
            y0, y1 = m(x[:32]), m(x[32:])
            whole = m(x)
        finally:
            E.set_option("two_streams", True)
        torch.cuda.synchronize()
        alone = torch.cat((y0, y1), 0)
        if prec == "fp32":
            assert torch.equal(y, alone)
        else:
            sc = max(1.0, float(alone.float().abs().max()))
            assert float((y.float() - alone.float()).abs().max()) <= 3e-3 * sc
        tol = 1e-4 if prec == "fp32" else 3e-3
        scale = max(1.0, float(whole.float().abs().max()))
        assert float((y.float() - whole.float()).abs().max()) <= tol * scale
    finally:
        tlxcv_amd.set_precision("fp32")


@pytest.mark.gpu
def test_two_streams_inside_a_captured_graph():
    from tlxcv_amd.graph import GraphedForward
    dev = torch.device("cuda:0")
    tlxcv_amd.set_precision("fp16")
    try:
        m = models.resnet50()
        m.load_dict(seeded.fill(seeded.shapes_of(m), 1))
        m = m.to(dev).set_eval()
        x = torch.from_numpy(seeded.image_batch(32, 0)).to(dev).repeat(4, 1, 1, 1).contiguous()
        want = m(x).clone()
        g = GraphedForward(m, x.clone(), warmup=2)
        got = g(x).clone()
        got2 = g(x.flip(0).contiguous()).clone()
        torch.cuda.synchronize()
        assert torch.equal(got, want)
        assert torch.equal(got2, want.flip(0))
    finally:
        tlxcv_amd.set_precision("fp32")


@pytest.mark.gpu
def test_planning_hint_changes_tiles_not_results():
    """TLXMI_PLAN_SHARED_HALF in the descriptor's flags (include/tlxmi.h): a layer planned for half the CUs gives the same
    values to fp16 rounding (another tile shape, another summation order at most); the hint is per call, nothing lingers."""
    dev = torch.device("cuda:0")
    tlxcv_amd.set_precision("fp16")
    try:
        g = torch.Generator().manual_seed(3)
        x = torch.randn((32, 14, 14, 256), generator=g).half().to(dev)
        pk = E.PackedFilter((torch.randn((256, 256, 3, 3), generator=g) * (2 / 2304) ** 0.5).to(dev), torch.float16)
        assert E.plan_flags() == 0
        y_full = E.conv2d(x, pk, 1, 1, 1, None, None, None, E.ACT_RELU)
        with E.shared_plan("half"):
            assert E.plan_flags() == 0x100
            y_half = E.conv2d(x, pk, 1, 1, 1, None, None, None, E.ACT_RELU)
        assert E.plan_flags() == 0
        y_again = E.conv2d(x, pk, 1, 1, 1, None, None, None, E.ACT_RELU)
        torch.cuda.synchronize()
        scale = float(y_full.float().abs().max())
        assert float((y_full.float() - y_half.float()).abs().max()) <= 2e-3 * scale
        assert torch.equal(y_full, y_again)
    finally:
        tlxcv_amd.set_precision("fp32")


@pytest.mark.gpu
def test_two_host_threads_enqueue_forwards_without_sharing_planning_state():
    """tile planning used to be a process-global flipped around every two-stream forward.  Two host
    threads, each with its own model, stream and batch — one of them inside a two-stream forward most of the time — must
    give what each gives alone, bit for bit (fp32), and the hint of one thread must never show up in the other."""
    import threading
    from tlxcv_amd import models, seeded
    dev = torch.device("cuda:0")
    tlxcv_amd.set_precision("fp32")
    try:
        ms, xs, wants = [], [], []
        for seed, batch in ((1, 8), (2, 6)):
            m = models.resnet18()
            m.load_dict(seeded.fill(seeded.shapes_of(m), seed))
            m = m.to(dev).set_eval()
            x = torch.from_numpy(seeded.image_batch(batch, seed, hw=64)).to(dev)
            ms.append(m); xs.append(x); wants.append(m(x).clone())
        torch.cuda.synchronize()
        errs, seen = [], []

        def work(i):
            try:
                st = torch.cuda.Stream(device=dev)
                with torch.cuda.stream(st):
                    for it in range(6):
                        if i == 0:
                            y = E.run_halves(lambda h: ms[0](h), xs[0], "half")      # the hint is live on THIS thread only
                        else:
                            seen.append(E.plan_flags())
                            y = ms[1](xs[1])
                        st.synchronize()
                        if not torch.equal(y, wants[i]):
                            errs.append((i, it, float((y - wants[i]).abs().max())))
            except Exception as e:      # noqa: BLE001
                errs.append((i, repr(e)))

        ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert not errs, errs
        assert seen and all(v == 0 for v in seen)
    finally:
        tlxcv_amd.set_precision("fp32")
