"""Coefficient tables of Pillow's two-pass resampler (the host pipeline's `Resize`, which goes through PIL.Image.resize),
restated from its published algorithm (Pillow, src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
ImagingResampleHorizontal_8bpc / Vertical_8bpc), so that the device kernels (tlxmi_resize_u8 ...) reproduce the host
result bit for bit: per axis, output sample xx takes the `count` input samples from `xmin` on with integer weights
kk (22 fractional bits), sum starts at 1 << 21, result = clip(sum >> 22, 0, 255); horizontal pass first, its uint8
result is the vertical pass's input.  Only what uint8 images need."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2

_FILTERS = {
    "bilinear": (1.0, lambda x: np.where(np.abs(x) < 1.0, 1.0 - np.abs(x), 0.0)),
    "nearest": None,
}


def _bicubic(x, a=-0.5):
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


_FILTERS["bicubic"] = (2.0, _bicubic)


def coefficients(in_size, out_size, interpolation="bilinear"):
    """-> (bounds int32 [out_size][2] = (xmin, count), kk int32 [out_size][ksize]) for one axis, full-image box."""
    support0, filt = _FILTERS[interpolation]
    scale = float(in_size) / float(out_size)
    filterscale = scale if scale >= 1.0 else 1.0
    support = support0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = filt((np.arange(xmax, dtype=np.float64) + xmin - center + 0.5) * ss).astype(np.float64)
        ww = 0.0
        for v in w:                     # the C code accumulates in this order
            ww += v
        if ww != 0.0:
            w = w / ww
        q = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)), (0.5 + w * (1 << PRECISION_BITS)))
        kk[xx, :xmax] = q.astype(np.int64).astype(np.int32)       # C (int) cast: truncation toward zero
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def resize_u8_numpy(img, size, interpolation="bilinear"):
    """Pure-numpy run of the same two passes (test oracle for the tables; the product path is the HIP kernel)."""
    a = np.asarray(img)
    assert a.dtype == np.uint8 and a.ndim == 3
    oh, ow = size
    H, W, _ = a.shape

    def one_axis(src, n_in, n_out, axis):
        if n_in == n_out:
            return src
        b, kk = coefficients(n_in, n_out, interpolation)
        src = np.moveaxis(src, axis, 0).astype(np.int64)
        out = np.empty((n_out,) + src.shape[1:], dtype=np.uint8)
        for xx in range(n_out):
            x0, n = b[xx]
            acc = np.tensordot(kk[xx, :n].astype(np.int64), src[x0:x0 + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
            out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
        return np.moveaxis(out, 0, axis)
    return one_axis(one_axis(a, W, ow, 1), H, oh, 0)
