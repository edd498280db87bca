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


# ---- classifier heads: few rows, large filter -> K slices side by side + deterministic reduction (tlxmi_linear_splitk, engine._linear_splits;
# vgg.py:36-60, alexnet.py:50-60, resnet.py:234-237 — round 4 widened the rule to the ResNet 2048 -> 1000 heads at <= 512 rows)
@pytest.mark.parametrize("dtype", [torch.float16, torch.float32], ids=["fp16", "fp32"])
@pytest.mark.parametrize("shape,act,res", [((64, 25088, 512), E.ACT_RELU, False), ((256, 4096, 1000), E.ACT_NONE, False),
                                           ((4, 4096, 4096), E.ACT_RELU, False), ((130, 9216, 520), E.ACT_NONE, True),
                                           ((1, 8192, 1000), E.ACT_NONE, False), ((256, 2048, 1000), E.ACT_NONE, False),
                                           ((128, 2048, 1000), E.ACT_NONE, False)], ids=lambda v: str(v).replace(" ", ""))
def test_linear_splitk(dev, dtype, shape, act, res):
    M, K, Cout = shape
    rng = np.random.default_rng(71)
    x = rnd(rng, (M, K))
    w = rnd(rng, (Cout, K), (1.0 / K) ** 0.5)
    b = rnd(rng, (Cout,), 0.2)
    r = rnd(rng, (M, Cout)) if res else None
    if dtype == torch.float16:
        x, w = q16(x), q16(w)
        r = q16(r) if r is not None else None
    want = x @ w.t() + b
    if r is not None:
        want = want + r
    if act == E.ACT_RELU:
        want = torch.relu(want)
    pk = E.PackedFilter(w.to(dev), dtype)
    xd = x.to(dtype).to(dev)
    assert E._linear_splits(M, K, pk, xd) >= 2          # the shapes above take the split path
    got = E.linear(xd, pk, b.to(dev), r.to(dtype).to(dev) if r is not None else None, act)
    got2 = E.linear(xd, pk, b.to(dev), r.to(dtype).to(dev) if r is not None else None, act)
    torch.cuda.synchronize()
    assert torch.equal(got, got2)                         # fixed summation order
    # partial sums are fp32 in both modes: the only fp16 rounding is the final store (half an ulp of the result)
    t = dict(atol=1e-4, rtol=1e-4) if dtype == torch.float32 else dict(atol=1e-3, rtol=1e-3)
    torch.testing.assert_close(got.float().cpu(), want, **t)
    if dtype == torch.float16:
        # the split path equals the unsplit GEMM (fp32 accumulate, one rounding) to fp16 rounding of the result
        E.set_option("splitk", False)
        try:
            plain = E.linear(xd, pk, b.to(dev), r.to(dtype).to(dev) if r is not None else None, act)
        finally:
            E.set_option("splitk", True)
        torch.testing.assert_close(got.float().cpu(), plain.float().cpu(), atol=1e-3, rtol=1e-3)


# ---- LayerNorm folded AROUND the Linear layers (round 5: tlxmi_linear_stats -> tlxmi_linear_ln, no launch in between; reference
# vision_transformer.py:172-175: x = x + attn(norm1(x)); x = x + mlp(norm2(x))).  The producer's outputs must equal the plain Linear's
# bit for bit, its row statistics the LayerNorm's, and the consumer the oracle's Linear(LayerNorm(x)) — on ragged row counts, several
# tiles per workgroup, the half-height tail tiles, a planned half device, rows far from zero mean, gamma / beta / bias present.
LN_CASES = [(2304, 768, 768, 2304), (12544, 512, 512, 2048), (3136 + 9, 1024, 1024, 3072), (25216, 768, 768, 3072), (25216 + 77, 3072, 768, 2304), (12544, 2048, 512, 1536), (50432, 768, 768, 768),
            (4096 + 5, 1408, 1024, 4096)]


@pytest.mark.parametrize("plan", [None, "half"], ids=["device", "half_device"])
@pytest.mark.parametrize("case", LN_CASES, ids=lambda c: "x".join(map(str, c)))
def test_layernorm_folded_around_linears(dev, case, plan):
    M, K, D, N2 = case
    rng = np.random.default_rng(M % 89)
    x = q16(rnd(rng, (M, K)))
    w = q16(rnd(rng, (D, K), (1.0 / K) ** 0.5))
    b = rnd(rng, (D,), 0.2)
    r = q16(rnd(rng, (M, D)))
    r[5] += 40.0                 # rows far from zero mean: the variance must not cancel away
    r[M - 3] -= 25.0
    r[M // 2, :32] *= 30.0       # one 32-channel slot that dominates its row
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, D).astype(np.float32))
    beta = rnd(rng, (D,), 0.3)
    w2 = rnd(rng, (N2, D), (1.0 / D) ** 0.5)
    b2 = rnd(rng, (N2,), 0.2)
    eps = 1e-6
    y_ref = x @ w.t() + b + r                                       # fp32
    xd, rd = x.half().to(dev), r.half().to(dev)
    pk = E.PackedFilter(w.to(dev), torch.float16)
    with E.shared_plan(plan):
        plain = E.linear(xd, pk, b.to(dev), res=rd)
        y, part = E.linear_stats(xd, pk, b.to(dev), res=rd)
        y2, part2 = E.linear_stats(xd, pk, b.to(dev), res=rd)
        prep = E.LinearLN(w2.to(dev), b2.to(dev), gamma.to(dev), beta.to(dev), torch.float16)
        outs = {act: E.linear_ln(y, prep, part, eps, act) for act in (E.ACT_NONE, E.ACT_GELU)}
        # without a residual (the patch-embedding producer of a model whose position table is folded elsewhere)
        y0, part0 = E.linear_stats(xd, pk, b.to(dev))
    torch.cuda.synchronize()
    assert torch.equal(y, y2) and torch.equal(part[:, :D // 256], part2[:, :D // 256])          # bit-reproducible
    # the plain Linear of the dispatcher (possibly other tiles, another summation order): equal to rounding
    torch.testing.assert_close(y.float(), plain.float(), atol=4e-3, rtol=4e-3)
    torch.testing.assert_close(y.float().cpu(), y_ref, **tol(torch.float16))
    torch.testing.assert_close(y0.float().cpu(), x @ w.t() + b, **tol(torch.float16))
    # row statistics: (sum, sum of squares) of the fp32 values before the store's rounding, per 256-channel tile column
    T = D // 256
    assert part.shape == (M, 4, 2) and part0.shape == part.shape
    planes = y_ref.view(M, T, 256)
    want_s, want_q = planes.sum(-1), (planes * planes).sum(-1)
    sc = float(y_ref.abs().max())
    torch.testing.assert_close(part[:, :T, 0].cpu(), want_s, atol=4e-3 * sc, rtol=2e-3)
    torch.testing.assert_close(part[:, :T, 1].cpu(), want_q, atol=4e-3 * sc * sc, rtol=4e-3)
    s0 = (x @ w.t() + b).view(M, T, 256)
    torch.testing.assert_close(part0[:, :T, 0].cpu(), s0.sum(-1), atol=4e-3 * sc, rtol=2e-3)
    yf = y.float().cpu()
    # consumer: Linear(LayerNorm(y)) of the oracle on the stored fp16 rows
    ln = OF.layernorm({"n.gamma": gamma, "n.beta": beta}, "n", yf, eps)
    for act, got in outs.items():
        want = ln @ w2.t() + b2
        if act == E.ACT_GELU:
            want = torch.nn.functional.gelu(want)
        # LN output O(1), K = D terms: the folded form rounds W * gamma instead of LN(y); allow a few fp16 ulps of the O(1) sum
        torch.testing.assert_close(got.float().cpu(), want, atol=6e-3, rtol=6e-3)


def test_folded_layernorm_refuses_what_the_persistent_kernel_does_not_take(dev):
    """tlxmi_linear_ln_supported and the UNSUPPORTED returns agree: fp32, Cout % 32, short K with a residual, an activation on the producer."""
    lib = _load()
    assert lib.tlxmi_linear_ln_supported(0, 25216, 768, 768, 0, 1) == 1
    assert lib.tlxmi_linear_ln_supported(0, 25216, 512, 512, 0, 1) == 1          # 8 K tiles: the one-tile-per-workgroup kernel (gemm_pp LNF)
    assert lib.tlxmi_linear_ln_supported(0, 25216, 512, 1536, 0, 0) == 1
    assert lib.tlxmi_linear_ln_supported(1, 25216, 768, 768, 0, 0) == 0          # fp32: the stand-alone LayerNorm keeps the reference's order
    assert lib.tlxmi_linear_ln_supported(0, 25216, 768, 200, 0, 0) == 0
    assert lib.tlxmi_linear_ln_supported(0, 25216, 768, 3072, E.ACT_GELU, 0) == 1
    assert lib.tlxmi_linear_ln_supported(0, 25216, 768, 3072, E.ACT_RELU, 0) == 0
    x = torch.zeros((512, 200), dtype=torch.float16, device=dev)
    pk = E.PackedFilter(torch.zeros((200, 200), device=dev), torch.float16)
    with pytest.raises(RuntimeError, match="outside"):
        E.linear_stats(x, pk, None)


def _load():
    from tlxcv_amd import _lib
    return _lib.load()


@pytest.mark.parametrize("C", [256, 512, 768, 1024])
def test_linear_ln_row_statistics_envelope(dev, C):
    """The consumer's own mean / rstd (from 1 .. 4 planes of (sum, sum of squares)): rows whose mean is 50 sigma away from zero, rows of
    tiny and of large variance, a ragged row count — against Linear(LayerNorm(x)) of the oracle in float64 on the same fp16 rows.
    (sum, sum of squares) in fp32 carry the variance to about 1e-7 * (1 + mean^2 / var) relative: 2.5e-4 here — the format's limit.)"""
    rng = np.random.default_rng(C)
    rows, N = 2304 + 37, 512
    x = rnd(rng, (rows, C))
    x[3] += 50.0
    x[rows - 1] = x[rows - 1] * 0.1 - 5.0
    x[7] *= 1e-2
    x[11] *= 30.0
    x = q16(x)
    planes = x.double().view(rows, C // 256, 256)
    part = torch.full((rows, 4, 2), float("nan"))                   # the pairs past C / 256 are never written by a producer: must not be read
    part[:, :C // 256] = torch.stack([planes.sum(-1), (planes * planes).sum(-1)], -1).float()
    part = part.to(dev)
    w = rnd(rng, (N, C), (1.0 / C) ** 0.5)
    b = rnd(rng, (N,), 0.2)
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32))
    beta = rnd(rng, (C,), 0.3)
    eps = 1e-5
    prep = E.LinearLN(w.to(dev), b.to(dev), gamma.to(dev), beta.to(dev), torch.float16)
    got = E.linear_ln(x.half().to(dev), prep, part, eps).float().cpu()
    xd = x.double()
    ln = (xd - xd.mean(1, keepdim=True)) / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + eps) * gamma.double() + beta.double()
    want = (ln @ w.double().t() + b.double()).float()
    torch.testing.assert_close(got, want, atol=8e-3, rtol=8e-3)
    with pytest.raises(RuntimeError, match="statistics of shape"):
        E.linear_ln(x.half().to(dev), prep, part[:-1].contiguous(), eps)


@pytest.mark.parametrize("shape", [(2304, 768, 2304), (2304, 768, 256), (27648, 768, 2304), (2304, 768, 3072), (12544 + 100, 512, 1536)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("act", [E.ACT_NONE, E.ACT_GELU], ids=["none", "gelu"])
def test_linear_ln_row_and_channel_tables_exact(dev, shape, act):
    """The regression test of the round-5 fault in gemm_stream's ROWAFF epilogue: with x = 0 and row statistics that make (a, b) = (1, 1)
    exactly (mean -1, variance 1, eps 0, spread over the planes) the consumer's output is act(c1[n] + c2[n]) — small integers, exact
    in fp16 — so ANY wrong read of a row's (a, b), of a plane, or of the channel tables (c1, c2) shows as a wrong integer.  (The
    first form read (a, b) of a sub-tile by one ds_read_b64 right in front of its use inside the MFMA segment: sub-tile 1 of the lower
    half tile came back with b = 0 in lanes 48 - 63 for single elements, on every launch, and only there — 36 of 2304 columns in 32
    of every 256 rows; the shipped form fetches the four pairs of a half tile in the load segment by one asm block.)  One and several
    tiles per workgroup, the half-height tail, both activations."""
    M, K, N = shape
    T = K // 256
    rng = np.random.default_rng(M + N)
    w = rnd(rng, (N, K), (1.0 / K) ** 0.5)
    prep = E.LinearLN(w.to(dev), None, torch.ones(K, device=dev), torch.zeros(K, device=dev), torch.float16)
    prep.c1 = (torch.arange(N, device=dev) % 500 + 1).float()
    prep.c2 = torch.full((N,), -7.0, device=dev)
    zeros = torch.zeros((M, K), dtype=torch.float16, device=dev)
    part = torch.full((M, 4, 2), float("nan"), device=dev)
    part[:, :T, 0] = -K / T                    # mean -1
    part[:, :T, 1] = 2.0 * K / T               # E[x^2] = 2 -> variance 1 -> rstd 1, b = -mean * rstd = 1
    want = prep.c1 - 7.0
    if act == E.ACT_GELU:
        want = torch.nn.functional.gelu(want)
    for plan in (None, "half"):
        with E.shared_plan(plan):
            z = E.linear_ln(zeros, prep, part, 0.0, act).float()
        bad = torch.nonzero((z - want[None]).abs() > (0 if act == E.ACT_NONE else 2e-3 * (1 + want.abs().max())))
        assert bad.shape[0] == 0, f"plan {plan}: {bad.shape[0]} wrong elements, first {bad[:6].tolist()}, rows mod 256 {sorted(set((bad[:, 0] % 256).tolist()))[:16]}"
    # and the row side: mean 0, rstd = a row pattern r[m] (variance 1 / r^2, only the LAST plane non-zero): out[m][n] = r[m] * (x W'^T)[m][n] + c2
    x = rnd(rng, (M, K)).half().to(dev)
    r = (torch.arange(M, device=dev) % 13 + 1).float()
    one = torch.zeros((M, 4, 2), device=dev)
    one[:, T - 1, 1] = K
    pat = torch.zeros((M, 4, 2), device=dev)
    pat[:, T - 1, 1] = K / (r * r)
    z0 = E.linear_ln(x, prep, one, 0.0, E.ACT_NONE).float()
    z2 = E.linear_ln(x, prep, pat, 0.0, E.ACT_NONE).float()
    torch.testing.assert_close(z2 + 7.0, (z0 + 7.0) * r[:, None], atol=8e-2, rtol=5e-3)      # (both sides rounded to fp16 at |z| up to ~40)
