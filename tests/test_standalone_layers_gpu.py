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
