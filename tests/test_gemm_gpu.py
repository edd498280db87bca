"""The 256 x 256 GEMM kernels behind tlxmi_conv2d (gemm_pp.hip = candidate 7, gemm_stream.hip = candidate 8, gemm_w4.hip = candidate 11;
Linear layers of reference vision_transformer.py:81-87,112-123) against the CPU oracle, one candidate
forced at a time through TLXMI_TILE — the dispatcher would otherwise pick them only for large layers.

Edge cases of the tiling and of the persistent K-tile stream: row tails (M % 256), channel tails
(Cout % 256, Cout not a multiple of 256 at all), 2 / 3 / odd / many K tiles, a K tail inside the last
128-byte tile, more tiles than CUs (several tiles per workgroup, unequal counts), fewer tiles than CUs,
bias / BatchNorm scale / residual / activation combinations, and the tail split of the dispatcher.
"""
import os

import numpy as np
import pytest
import torch

from oracle import functional as OF
from tlxcv_amd import engine as E
from util import rnd, q16, tol

pytestmark = pytest.mark.gpu


@pytest.fixture
def force_tile():
    """set_tile(t): the following launches run on the TUNING flavour of the library (libtlxmi_tune.so, which reads
    TLXMI_TILE per call) with tile candidate t forced; set_tile(None): back to the product library and its own choice."""
    from tlxcv_amd._lib import tuning
    state = {"ctx": None}

    def set_tile(t):
        if state["ctx"] is not None:
            state["ctx"].__exit__(None, None, None)
            state["ctx"] = None
        if t is not None:
            state["ctx"] = tuning(TLXMI_TILE=str(t))
            state["ctx"].__enter__()
    yield set_tile
    set_tile(None)


def run_linear(dev, dtype, M, K, Cout, bias=True, scale=False, res=False, act=E.ACT_NONE, seed=0):
    rng = np.random.default_rng(seed)
    x = rnd(rng, (M, K))
    w = rnd(rng, (Cout, K), (1.0 / K) ** 0.5)
    b = rnd(rng, (Cout,), 0.2) if bias else None
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, Cout).astype(np.float32)) if scale else None
    r = rnd(rng, (M, Cout)) if res else None
    if dtype == torch.float16:
        x, w = q16(x), q16(w)
        r = q16(r) if r is not None else None
    # oracle: the 1x1 convolution restatement with M as the pixel axis
    want = OF.conv_bn_act(x.t().reshape(1, K, M, 1), w.reshape(Cout, K, 1, 1), sc, b,
                          r.t().reshape(1, Cout, M, 1) if r is not None else None, act, 0.0, (1, 1), (0, 0), 1, 1, False)
    want = want.reshape(Cout, M).t()
    pk = E.PackedFilter(w.reshape(Cout, K, 1, 1).to(dev), dtype)
    xe = x.to(dtype).to(dev).view(M, 1, 1, K)
    re_ = r.to(dtype).to(dev).view(M, 1, 1, Cout) if r is not None else None
    got = E.conv2d(xe, pk, 1, 0, 1, sc.to(dev) if sc is not None else None, b.to(dev) if b is not None else None, re_, act)
    torch.cuda.synchronize()
    torch.testing.assert_close(got.view(M, Cout).float().cpu(), want, **tol(dtype))


# (M, K, Cout): K in elements
SHAPES = [
    (256, 128, 256),        # one tile, 2 K tiles (fp16) — fewer tiles than CUs
    (300, 192, 256),        # row tail, 3 K tiles
    (1000, 200, 264),       # row tail, channel tail (Cout % 256 = 8), K tail inside the last tile
    (513, 448, 512),        # 7 K tiles (fp16), 3 x 2 tiles
    (3 * 256, 768, 768),    # ViT proj shape at small M: 12 K tiles (residual path of the stream kernel)
    (70 * 256 + 17, 128, 1024),   # 284 tiles > 256 CUs: two tiles for some workgroups, one for the rest
    (256 * 20, 704, 3072),  # 240 tiles, 11 K tiles (smallest residual-capable stream)
]


@pytest.mark.parametrize("tile", [7, 8, 11], ids=["pp", "stream", "w4"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.float32], ids=["fp16", "fp32"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_linear_bias(dev, force_tile, tile, dtype, shape):
    force_tile(tile)
    run_linear(dev, dtype, *shape)


@pytest.mark.parametrize("tile", [7, 8, 11], ids=["pp", "stream", "w4"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.float32], ids=["fp16", "fp32"])
@pytest.mark.parametrize("epi", ["none", "gelu", "relu", "res", "res_relu", "bn_relu", "bn_res", "nobias_res"])
def test_linear_epilogues(dev, force_tile, tile, dtype, epi):
    force_tile(tile)
    kw = dict(bias=epi != "nobias_res", scale=epi.startswith("bn"), res="res" in epi,
              act=E.ACT_GELU if epi == "gelu" else E.ACT_RELU if "relu" in epi else E.ACT_NONE)
    run_linear(dev, dtype, 2 * 256 + 40, 768, 768, seed=3, **kw)


@pytest.mark.parametrize("tile", [7, 8, 11], ids=["pp", "stream", "w4"])
def test_many_tiles_per_workgroup(dev, force_tile, tile):
    # 197 x 3 tiles of the ViT-B proj / fc2 layers at batch 256: 2-3 tiles per workgroup with a residual
    force_tile(tile)
    run_linear(dev, torch.float16, 50432, 768, 768, res=True, seed=5)


TAIL_CASES = [(50432, 768, 768, True, E.ACT_NONE), (25216, 768, 768, True, E.ACT_NONE), (25216, 3072, 768, True, E.ACT_NONE),
              (25216 + 77, 768, 2304, False, E.ACT_NONE), (12544, 512, 512, True, E.ACT_NONE), (12544, 2048, 512, False, E.ACT_RELU),
              (197 * 3 + 5, 768, 768, True, E.ACT_NONE), (50432, 3072, 768, False, E.ACT_GELU)]


@pytest.mark.parametrize("plan", [None, "half"], ids=["device", "half_device"])
@pytest.mark.parametrize("case", TAIL_CASES, ids=lambda c: "x".join(map(str, c[:3])) + ("+res" if c[3] else ""))
def test_balanced_tails_of_the_persistent_gemm(dev, force_tile, case, plan):
    """A last round of 256 x 256 tiles that is at most half full (ViT-B/16 proj / fc2: 591 tiles on 256 CUs, 297 on the 128 a
    two-stream forward plans for), three ways, all against the oracle and against each other:
      (a) as it was: the short round on whole tiles (tail_splitk off, TLXMI_HALFTAIL=0);
      (b) gemm_stream.hip's half-height tiles inside the persistent kernel (tail_splitk off) — bit-identical to (a) without a
          residual (same K order per output; with one, a row's residual enters its fp32 sum at another K tile);
      (c) engine._linear_tail: whole rounds on the persistent kernel + the leftover rows on K slices with the slice-order
          reduction (option "tail_splitk": measured a loss on the forward, off by default) — another summation order, equal to
          rounding, bit-reproducible."""
    from tlxcv_amd._lib import tuning
    M, K, Cout, res, act = case
    force_tile(8)
    with E.shared_plan(plan):
        run_linear(dev, torch.float16, M, K, Cout, res=res, act=act, seed=M % 97)      # conv2d entry: (b) where it applies
        rng = np.random.default_rng(3)
        xc = q16(rnd(rng, (M, K)))
        wc = q16(rnd(rng, (Cout, K), (1.0 / K) ** 0.5))
        bc = rnd(rng, (Cout,), 0.2)
        rc = q16(rnd(rng, (M, Cout))) if res else None
        x, b = xc.half().to(dev), bc.to(dev)
        pk = E.PackedFilter(wc.to(dev), torch.float16)
        r = rc.half().to(dev) if res else None
        E.set_option("tail_splitk", True)
        try:
            y_c = E.linear(x, pk, b, res=r, act=act)                                      # (c) where it applies
            assert torch.equal(E.linear(x, pk, b, res=r, act=act), y_c)
        finally:
            E.set_option("tail_splitk", False)
        y_b = E.linear(x, pk, b, res=r, act=act)
        with tuning(TLXMI_TILE="8", TLXMI_HALFTAIL="0"):
            y_a = E.linear(x, pk, b, res=r, act=act)
        torch.cuda.synchronize()
        if not res:
            assert torch.equal(y_a, y_b)
        for y in (y_b, y_c):
            torch.testing.assert_close(y.float(), y_a.float(), atol=4e-3, rtol=4e-3)
            assert (y != y_a).float().mean().item() < 0.05
        # (c) against the oracle on the rows of the K slices (the last rows) and on the first rows
        sel = torch.cat([torch.arange(0, min(M, 300)), torch.arange(max(0, M - 700), M)]).unique()
        want = xc[sel] @ wc.t() + bc
        if res:
            want = want + rc[sel]
        want = torch.nn.functional.gelu(want) if act == E.ACT_GELU else torch.relu(want) if act == E.ACT_RELU else want
        torch.testing.assert_close(y_c[sel.to(dev)].float().cpu(), want, **tol(torch.float16))


def test_auto_dispatch_tail_on_half_height_tiles(dev, force_tile):
    # 197 x 3 tiles = 2 rounds + 79: 170 row tiles go to the 256 x 256 kernel, the last 6912 rows to the same
    # antiphase kernel on 128 x 256 tiles (gemm_pp128), with bias + residual
    force_tile(None)
    run_linear(dev, torch.float16, 50432, 768, 768, res=True, seed=9)


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32], ids=["fp16", "fp32"])
def test_half_height_kernel_directly(dev, force_tile, dtype):
    # small enough that only the tail path's shape rules matter: 2 rounds + a tail needs > 512 tiles, so the
    # kernel is also reached through TLXMI_TILE=7 on a problem whose row count is not a multiple of 128
    force_tile(None)
    run_linear(dev, dtype, 256 * 170 * 1 + 128 * 3 + 5, 256, 768, act=E.ACT_RELU, scale=True, seed=10)


def test_auto_dispatch_tail_split(dev, force_tile):
    # 197 x 12 tiles = 9 rounds + 60: the dispatcher sends 192 row tiles to the 256 x 256 kernel and the last
    # 1280 rows to small tiles; the seam must be invisible
    force_tile(None)
    run_linear(dev, torch.float16, 50432, 768, 3072, act=E.ACT_GELU, seed=7)


# ---- K = 128 / 256 on many rows: the filter-in-registers streaming kernel (gemm_wreg.hip; Swin-B stages 1 and 2,
# swin_transformer.py:192-229, 37-50): every compiled width (K = 128: 128 / 256 / 384 / 512 output channels; K = 256: 256 / 512 / 768 /
# 1024, the last two as two column slices), a ragged last row tile, every epilogue family (the residual ones stay on the tiled
# kernels), against the oracle and against the tiled kernels (TLXMI_WREG=0) on the same inputs
@pytest.mark.parametrize("epi", ["bias", "gelu", "res", "bn_relu_res", "nobias"])
@pytest.mark.parametrize("K,Cout", [(128, 128), (128, 256), (128, 384), (128, 512), (256, 256), (256, 512), (256, 768), (256, 1024)])
def test_linear_k128_filter_in_registers(dev, K, Cout, epi):
    from tlxcv_amd._lib import tuning
    M = (16384 if K == 128 else 32768) + 16 * 5 + 3          # the last row tile ragged (19 / 83 rows)
    kw = dict(bias=epi != "nobias", scale=epi == "bn_relu_res", res=epi in ("res", "bn_relu_res"),
              act={"gelu": E.ACT_GELU, "bn_relu_res": E.ACT_RELU}.get(epi, E.ACT_NONE), seed=Cout)
    run_linear(dev, torch.float16, M, K, Cout, **kw)                      # the product's own choice of kernel, vs the oracle
    rng = np.random.default_rng(7)
    x = q16(rnd(rng, (M, K)))
    w = q16(rnd(rng, (Cout, K), (1.0 / K) ** 0.5))
    b = rnd(rng, (Cout,), 0.2)
    want = OF.conv_bn_act(x.t().reshape(1, K, M, 1), w.reshape(Cout, K, 1, 1), None, b, None, kw["act"], 0.0, (1, 1), (0, 0), 1, 1,
                          False).reshape(Cout, M).t()
    xe = x.half().to(dev).view(M, 1, 1, K)
    pk = E.PackedFilter(w.reshape(Cout, K, 1, 1).to(dev), torch.float16)
    with tuning(TLXMI_WREG="3"):             # every compiled width on the kernel under test (the product keeps 256 -> 256 / 512 tiled)
        got = E.conv2d(xe, pk, 1, 0, 1, None, b.to(dev), None, kw["act"])
    with tuning(TLXMI_WREG="0"):
        old = E.conv2d(xe, pk, 1, 0, 1, None, b.to(dev), None, kw["act"])
    torch.cuda.synchronize()
    torch.testing.assert_close(got.view(M, Cout).float().cpu(), want, **tol(torch.float16))
    torch.testing.assert_close(got.float(), old.float(), atol=2e-3, rtol=2e-3)
