"""Shared helpers for the GPU parity tests."""
import numpy as np
import torch


def rnd(rng, shape, scale=1.0):
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))


def q16(t):
    """Round to fp16 and back: the oracle then sees exactly the values the fp16 kernels read."""
    return t.half().float()


def nchw_to_engine(x_nchw, dtype, dev):
    """Reference-side layout helper (torch ops, test-only): NCHW fp32 -> NHWC engine tensor with the
    channel axis zero-padded to a whole number of 16-byte chunks."""
    v = 8 if dtype == torch.float16 else 4
    n, c, h, w = x_nchw.shape
    cp = (c + v - 1) // v * v
    out = torch.zeros((n, h, w, cp), dtype=dtype)
    out[..., :c] = x_nchw.permute(0, 2, 3, 1).to(dtype)
    return out.to(dev)


def engine_to_nchw(y_nhwc):
    return y_nhwc.float().cpu().permute(0, 3, 1, 2).contiguous()


def tol(dtype):
    # fp32: the north-star bound (1e-4 abs on O(1) values).  fp16: inputs are pre-rounded, the
    # kernel accumulates in fp32 and rounds once on store -> half an fp16 ulp relative (2^-11)
    # plus accumulation-order noise.
    return dict(atol=1e-4, rtol=1e-4) if dtype == torch.float32 else dict(atol=2e-3, rtol=2e-3)
