"""HIP implicit-GEMM conv / linear vs the CPU oracle, through the C-ABI (tlxmi_conv2d).

Every unique conv shape of ResNet-50 at batch 1-2, plus the edge cases the tiling
has: pixel-tile tails, channel tails that force the scalar store path (Cout 1000, 291, 3), padded
input channels (3 -> 8/4), dilation, asymmetric stride/padding, every epilogue combination.
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import functional as OF
from tlxcv_amd import engine as E, _lib
from tlxcv_amd._lib import tuning      # the tuning flavour of the library: honours TLXMI_TILE / TLXMI_HALO per call
from util import rnd, q16, nchw_to_engine, engine_to_nchw, tol

pytestmark = pytest.mark.gpu

# (Cin, Cout, k, stride, Hin) — resnet.py graph at 224 input, see SURVEY §8 a3
RESNET50_CONVS = [
    (3, 64, 7, 2, 224), (64, 64, 1, 1, 56), (64, 64, 3, 1, 56), (64, 256, 1, 1, 56), (256, 64, 1, 1, 56),
    (256, 128, 1, 1, 56), (128, 128, 3, 2, 56), (128, 512, 1, 1, 28), (256, 512, 1, 2, 56), (512, 128, 1, 1, 28),
    (128, 128, 3, 1, 28), (512, 256, 1, 1, 28), (256, 256, 3, 2, 28), (256, 1024, 1, 1, 14), (512, 1024, 1, 2, 28),
    (1024, 256, 1, 1, 14), (256, 256, 3, 1, 14), (1024, 512, 1, 1, 14), (512, 512, 3, 2, 14), (512, 2048, 1, 1, 7),
    (1024, 2048, 1, 2, 14), (2048, 512, 1, 1, 7), (512, 512, 3, 1, 7),
]


def run_case(dev, dtype, N, Cin, Cout, k, stride, pad, H, W=None, dil=1, act=0, act_param=0.0, with_bn=True,
             with_res=False, res_after=False, seed=0):
    W = W or H
    rng = np.random.default_rng(seed)
    kh, kw = (k, k) if isinstance(k, int) else k
    x = rnd(rng, (N, Cin, H, W))
    w = rnd(rng, (Cout, Cin, kh, kw), (2.0 / (Cin * kh * kw)) ** 0.5)
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, Cout).astype(np.float32)) if with_bn else None
    shift = rnd(rng, (Cout,), 0.1) if with_bn else None
    if dtype == torch.float16:
        x, w = q16(x), q16(w)
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    Ho = (H + 2 * ph - dil * (kh - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dil * (kw - 1) - 1) // sw + 1
    res = rnd(rng, (N, Cout, Ho, Wo)) if with_res else None
    if res is not None and dtype == torch.float16:
        res = q16(res)
    want = OF.conv_bn_act(x, w, scale, shift, res, act, act_param, (sh, sw), (ph, pw), dil, 1, res_after)

    pk = E.PackedFilter(w.to(dev), dtype)
    xe = nchw_to_engine(x, dtype, dev)
    re_ = res.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev) if res is not None else None
    got = E.conv2d(xe, pk, (sh, sw), (ph, pw), dil, scale.to(dev) if with_bn else None,
                   shift.to(dev) if with_bn else None, re_, act, act_param, res_after)
    torch.cuda.synchronize()
    assert got.shape == (N, Ho, Wo, Cout)
    torch.testing.assert_close(engine_to_nchw(got), want, **tol(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
@pytest.mark.parametrize("cfg", RESNET50_CONVS, ids=lambda c: "x".join(map(str, c)))
def test_resnet50_conv_shapes(dev, dtype, cfg):
    Cin, Cout, k, s, H = cfg
    run_case(dev, dtype, 2 if H <= 56 else 1, Cin, Cout, k, s, k // 2, H, act=1, with_res=(k == 1 and Cout >= 256))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
@pytest.mark.parametrize("act", list(range(9)))
def test_every_activation_epilogue(dev, dtype, act):
    run_case(dev, dtype, 1, 32, 48, 3, 1, 1, 9, act=act, act_param=0.1, with_res=True, res_after=(act == 3))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
@pytest.mark.parametrize("case", [
    dict(N=1, Cin=3, Cout=3, k=3, stride=1, pad=1, H=5),                      # tiny everything, scalar stores
    dict(N=3, Cin=16, Cout=291, k=1, stride=1, pad=0, H=13),                  # YOLO head width (yolov3.py:352)
    dict(N=5, Cin=2048, Cout=1000, k=1, stride=1, pad=0, H=1, with_bn=True),  # classifier GEMM
    dict(N=2, Cin=24, Cout=40, k=3, stride=1, pad=2, H=17, dil=2),            # dilation
    dict(N=1, Cin=8, Cout=72, k=(3, 5), stride=(2, 1), pad=(1, 2), H=19, W=23),  # asymmetric
    dict(N=2, Cin=3, Cout=768, k=16, stride=16, pad=0, H=64),                 # ViT patch embed geometry
    dict(N=1, Cin=3, Cout=128, k=4, stride=4, pad=0, H=56),                   # Swin patch embed geometry
    dict(N=1, Cin=64, Cout=64, k=1, stride=1, pad=0, H=1, with_bn=False),     # a single row
    dict(N=1, Cin=40, Cout=136, k=3, stride=2, pad=1, H=31, with_res=True),   # odd sizes + residual
], ids=lambda c: f"{c['Cin']}to{c['Cout']}k{c['k']}")
def test_edge_geometries(dev, dtype, case):
    run_case(dev, dtype, **case)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
def test_linear_with_bias_gelu_and_residual(dev, dtype):
    rng = np.random.default_rng(3)
    x, w, b, r = rnd(rng, (3, 50, 96)), rnd(rng, (96, 200), 0.1), rnd(rng, (200,), 0.1), rnd(rng, (3, 50, 200))
    if dtype == torch.float16:
        x, w, r = q16(x), q16(w), q16(r)
    pk = E.PackedFilter(w.t().contiguous().to(dev), dtype)
    got = E.linear(x.to(dtype).to(dev), pk, b.to(dev), act=E.ACT_GELU)
    want = torch.nn.functional.gelu(x @ w + b)
    torch.testing.assert_close(got.float().cpu(), want, **tol(dtype))
    got = E.linear(x.to(dtype).to(dev), pk, b.to(dev), res=r.to(dtype).to(dev))
    torch.testing.assert_close(got.float().cpu(), x @ w + b + r, **tol(dtype))


def test_linearity_at_full_size(dev):
    """Size-independent property at the BASELINE shape (bs 256, 56x56, 64->256, fp16): conv is linear,
    so conv(a) + conv(b) == conv(a + b) up to fp16 rounding, with no oracle in the loop."""
    rng = np.random.default_rng(1)
    g = torch.Generator(device="cpu").manual_seed(1)
    a = (torch.randn((256, 56, 56, 64), generator=g) * 0.5).half().to(dev)
    b = (torch.randn((256, 56, 56, 64), generator=g) * 0.5).half().to(dev)
    w = q16(rnd(rng, (256, 64, 1, 1), 0.15))
    pk = E.PackedFilter(w.to(dev), torch.float16)
    ya, yb, yab = E.conv2d(a, pk), E.conv2d(b, pk), E.conv2d((a.float() + b.float()).half(), pk)
    torch.cuda.synchronize()
    # a+b is rounded to fp16 before the conv: bound by K * ulp(a+b) * |w|
    err = (ya.float() + yb.float() - yab.float()).abs().max().item()
    assert err < 2e-2, err
    # and a spot check of 512 random output pixels against the oracle
    idx = torch.randint(0, 256 * 56 * 56, (512,), generator=g)
    xa = a.view(-1, 64)[idx.to(dev)].float().cpu()
    want = xa @ w.view(256, 64).t()
    torch.testing.assert_close(ya.view(-1, 256)[idx.to(dev)].float().cpu(), want, atol=2e-3, rtol=2e-3)


def test_descriptor_errors_raise(dev):
    pk = E.PackedFilter(torch.zeros(8, 8, 3, 3, device=dev), torch.float16)
    x = torch.zeros((1, 2, 2, 8), dtype=torch.float16, device=dev)
    with pytest.raises(RuntimeError, match="empty output"):
        E.conv2d(x, pk, 1, 0)                      # 3x3 valid on 2x2
    with pytest.raises(RuntimeError, match="dtype"):
        E.conv2d(x.float(), pk, 1, 1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
@pytest.mark.parametrize("k,s,p,b,H", [(7, 2, 3, 2, 64), (16, 16, 0, 4, 64), (4, 4, 0, 4, 56), (3, 2, 1, 2, 30),
                                       (16, 16, 0, 2, 32), (5, 2, 2, 2, 22)])
def test_space_to_depth_stem_equals_plain_conv(dev, dtype, k, s, p, b, H):
    """GroupConv2d.run_stem (b x b fold + re-indexed filter + cropped extent) == the plain conv."""
    import tlxcv_amd
    from tlxcv_amd.tlx import nn
    tlxcv_amd.set_precision("fp32" if dtype == torch.float32 else "fp16")
    try:
        rng = np.random.default_rng(5)
        conv = nn.GroupConv2d(in_channels=3, out_channels=40, kernel_size=k, stride=s, padding=p,
                              data_format="channels_first")
        w, bias = rnd(rng, (40, 3, k, k), 0.2), rnd(rng, (40,), 0.1)
        x = rnd(rng, (2, 3, H, H))
        if dtype == torch.float16:
            w, x = q16(w), q16(x)
        conv.load_dict({"filters": w, "biases": bias})
        conv = conv.to(dev).set_eval()
        want = torch.nn.functional.conv2d(x, w, bias, s, p)
        got = conv.run_stem(x.to(dev), b)
        torch.testing.assert_close(engine_to_nchw(got), want, **tol(dtype))
    finally:
        tlxcv_amd.set_precision("fp16")


# ---- conv_halo.hip (thin inputs, stride 1): forced on / off through TLXMI_HALO, both against the oracle
HALO_CASES = [
    # (N, Cin, Cout, k, pad, H, W, res, act)
    (2, 64, 64, 3, 1, 56, 56, False, 1),      # ResNet layer1 3x3 (resnet.py:142-156)
    (1, 64, 64, 3, 1, 56, 56, True, 1),       # BasicBlock conv2 + identity (resnet.py:90-106)
    (1, 64, 128, 3, 1, 40, 52, False, 0),     # two channel tiles, Wo does not divide 256
    (1, 64, 72, 3, 1, 33, 47, True, 3),       # channel tail, ragged last tile, leaky
    (1, 64, 64, 3, 0, 36, 36, False, 1),      # no padding
    (1, 32, 64, 3, 1, 64, 64, False, 3),      # 64 bytes per pixel (darknet.py:54-58)
    (2, 16, 64, 4, 0, 67, 67, False, 1),      # space-to-depth stem geometry: 4x4 taps over 16 channels
    (1, 16, 32, 2, 0, 65, 65, False, 2),      # 3x3 stride-2 stem after space-to-depth: 2x2 taps (mobilenetv1.py:79-88)
    (1, 12, 64, 4, 0, 115, 115, False, 1),    # ResNet stem exactly (12 real channels padded to 16)
]


@pytest.mark.parametrize("halo", ["1", "0"], ids=["halo", "igemm"])
@pytest.mark.parametrize("cfg", HALO_CASES, ids=lambda c: "x".join(map(str, c)))
def test_thin_input_stride1(dev, cfg, halo):
    N, Cin, Cout, k, pad, H, W, res, act = cfg
    with tuning(TLXMI_HALO=halo):
        run_case(dev, torch.float16, N, Cin, Cout, k, 1, pad, H, W, act=act, act_param=0.1, with_res=res, seed=11)


@pytest.mark.parametrize("cfg", [(128, 64, 64, 3, 1, 56, 56, True), (64, 16, 64, 4, 0, 115, 115, False)],
                         ids=["3x3x64_batch128", "stem_batch64"])
def test_thin_input_full_size_matches_the_implicit_gemm(dev, cfg):
    """BASELINE-sized property check: at full image count (dozens of tiles per workgroup, every ring wrap and
    image boundary of conv_halo.hip) the row-ring kernel and the implicit GEMM — each checked against the oracle
    at small sizes above — agree to fp16 rounding of one accumulation order vs the other."""
    N, Cin, Cout, k, pad, H, W, res = cfg
    g = torch.Generator().manual_seed(3)
    x = (torch.randn((N, H, W, Cin), generator=g) * 0.5).half().to(dev)
    w = torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    pk = E.PackedFilter(w.to(dev), torch.float16)
    sc = (torch.rand(Cout, generator=g) + 0.5).to(dev)
    sh = (torch.randn(Cout, generator=g) * 0.1).to(dev)
    Ho, Wo = H + 2 * pad - k + 1, W + 2 * pad - k + 1
    r = torch.randn((N, Ho, Wo, Cout), generator=g).half().to(dev) if res else None
    outs = []
    for halo in ("1", "0"):
        with tuning(TLXMI_HALO=halo):
            outs.append(E.conv2d(x, pk, 1, pad, 1, sc, sh, r, E.ACT_RELU).float().cpu())
    torch.testing.assert_close(outs[0], outs[1], atol=4e-3, rtol=4e-3)


# ---- grouped convolution (tlxmi_group_conv2d; resnext.py:83-91 cardinality 32 / 64): every ResNeXt-50 stage
# shape, odd group sizes that still merge into 16-byte chunks, residual + every chunk count, and the error path
GROUP_CASES = [
    # (N, Cin, Cout, groups, k, stride, pad, H, W, res, act)
    (2, 128, 128, 32, 3, 1, 1, 24, 24, False, 1),    # 32x4d stage 1: 4 channels per group, 2 launch chunks (fp16)
    (2, 256, 256, 32, 3, 2, 1, 23, 23, False, 1),    # stage 2 entry, stride 2, odd extent
    (1, 512, 512, 32, 3, 1, 1, 14, 14, False, 1),    # stage 3: 16 per group
    (1, 1024, 1024, 32, 3, 1, 1, 7, 7, False, 1),    # stage 4: 32 per group, 16 chunks
    (1, 256, 256, 64, 3, 1, 1, 12, 20, False, 1),    # 64x4d stage 1
    (1, 2048, 2048, 64, 3, 2, 1, 9, 9, False, 1),    # 64x4d stage 4 entry
    (1, 64, 128, 4, 3, 1, 1, 10, 10, True, 3),       # Cin != Cout per group (16 -> 32), residual, leaky
    (1, 96, 48, 3, 1, 1, 0, 6, 6, False, 0),         # 1x1, 3 groups of 32 -> 16 merged into one chunk
    (1, 48, 96, 2, 5, 1, 2, 11, 11, True, 2),        # 5x5, 2 groups of 24 -> 48
    (1, 40, 40, 5, 3, 1, 1, 8, 8, False, 1),         # 8 per group: groups merge until all 5 are one chunk
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
@pytest.mark.parametrize("cfg", GROUP_CASES, ids=lambda c: "x".join(map(str, c)))
def test_group_conv(dev, dtype, cfg):
    N, Cin, Cout, groups, k, stride, pad, H, W, with_res, act = cfg
    rng = np.random.default_rng(17)
    cg = Cin // groups
    x = rnd(rng, (N, Cin, H, W))
    w = rnd(rng, (Cout, cg, k, k), (2.0 / (cg * k * k)) ** 0.5)
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, Cout).astype(np.float32))
    shift = rnd(rng, (Cout,), 0.1)
    if dtype == torch.float16:
        x, w = q16(x), q16(w)
    Ho = (H + 2 * pad - k) // stride + 1
    Wo = (W + 2 * pad - k) // stride + 1
    res = rnd(rng, (N, Cout, Ho, Wo)) if with_res else None
    if res is not None and dtype == torch.float16:
        res = q16(res)
    want = OF.conv_bn_act(x, w, scale, shift, res, act, 0.1, (stride, stride), (pad, pad), 1, groups, False)
    pk = E.PackedGroupFilter(w.to(dev), groups, dtype)
    assert pk.chunks >= 1 and groups % pk.chunks == 0
    xe = x.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev)
    re_ = res.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev) if res is not None else None
    got = E.group_conv2d(xe, pk, stride, pad, 1, scale.to(dev), shift.to(dev), re_, act, 0.1)
    torch.cuda.synchronize()
    assert got.shape == (N, Ho, Wo, Cout)
    torch.testing.assert_close(engine_to_nchw(got), want, **tol(dtype))


# ---- the small-block MFMA kernel (group_conv.hip: v_mfma_f32_4x4x4_16b_f16, no products with the zero blocks of the
# block-diagonal filter) takes fp16 3x3 / padding 1 / stride 1 or 2 layers with 4, 8, 16 or 32 channels per group: every group
# width x stride, images taller than one LDS tile (several row tiles per image, a ragged last one), widths that are not a
# multiple of the 4-pixel run, a single column, every activation family, no BatchNorm — against the oracle AND against the
# block-diagonal path (TLXMI_GCONV=0 in the tuning flavour) on the same inputs.
GCONV_CASES = [
    # (N, C, groups, stride, H, W, act, with_bn)
    (2, 128, 32, 1, 56, 56, 1, True),      # resnext.py:83-91 stage 1 (32x4d): 14 row tiles of 4 rows
    (1, 128, 32, 2, 57, 55, 1, True),      # stride 2, odd extents
    (2, 256, 32, 1, 28, 28, 1, True),      # stage 2: 8 per group
    (1, 256, 32, 2, 56, 56, 3, True),      # stage 2 entry, leaky
    (3, 512, 32, 1, 14, 14, 1, True),      # stage 3: 16 per group
    (1, 512, 32, 2, 28, 28, 0, False),     # no BatchNorm, no activation
    (5, 1024, 32, 1, 7, 7, 1, True),       # stage 4: 32 per group; 49 pixels = 12 runs + 1
    (1, 1024, 32, 2, 14, 14, 4, True),     # hardswish
    (1, 256, 64, 1, 9, 13, 2, True),       # 64x4d, relu6, odd extents
    (2, 64, 16, 1, 5, 1, 1, True),         # one column
    (1, 192, 24, 2, 11, 6, 7, True),       # three chunks of 8 per group, silu
    (1, 128, 32, 1, 120, 90, 1, True),     # 92 padded columns: 4-row tiles, 30 of them
]


@pytest.mark.parametrize("cfg", GCONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_group_conv_on_the_small_block_mfma(dev, cfg):
    N, C, groups, stride, H, W, act, with_bn = cfg
    rng = np.random.default_rng(31 + C + H)
    cg = C // groups
    x = q16(rnd(rng, (N, C, H, W)))
    w = q16(rnd(rng, (C, cg, 3, 3), (2.0 / (cg * 9)) ** 0.5))
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32)) if with_bn else None
    shift = rnd(rng, (C,), 0.1) if with_bn else None
    want = OF.conv_bn_act(x, w, scale, shift, None, act, 0.1, (stride, stride), (1, 1), 1, groups, False)
    pk = E.PackedGroupFilter(w.to(dev), groups, torch.float16)
    xe = x.permute(0, 2, 3, 1).contiguous().half().to(dev)
    sc, sh = (scale.to(dev) if with_bn else None), (shift.to(dev) if with_bn else None)
    d = _lib.ConvDesc(dtype=E.dt_code(torch.float16), N=N, H=H, W=W, C=C, Cout=C, R=3, S=3, stride_h=stride, stride_w=stride, pad_h=1, pad_w=1,
                      dil_h=1, dil_w=1, Ho=(H - 1) // stride + 1, Wo=(W - 1) // stride + 1, x_ld=C, y_ld=C, res_ld=0, y_nstride=0,
                      res_nstride=0, act=act, act_param=0.1, flags=0)
    assert _lib.load().tlxmi_group_conv2d_small_supported(ctypes.byref(d), groups) == 1       # the kernel under test is the one that runs
    with tuning(TLXMI_GCONV="255"):          # every supported layer on the kernel under test (the product picks it where it is faster)
        got = E.group_conv2d(xe, pk, stride, 1, 1, sc, sh, None, act, 0.1)
    with tuning(TLXMI_GCONV="0"):
        old = E.group_conv2d(xe, pk, stride, 1, 1, sc, sh, None, act, 0.1)
    dflt = E.group_conv2d(xe, pk, stride, 1, 1, sc, sh, None, act, 0.1)     # the product library's own choice
    torch.cuda.synchronize()
    torch.testing.assert_close(engine_to_nchw(got), want, **tol(torch.float16))
    torch.testing.assert_close(engine_to_nchw(dflt), want, **tol(torch.float16))
    torch.testing.assert_close(got.float(), old.float(), atol=4e-3, rtol=4e-3)       # same fp32 accumulation of fp16 products, another order


def test_group_conv_rejects_unmergeable_channel_counts(dev):
    """3 channels per group, 5 groups: no merge of whole groups makes 16-byte pixel chunks (fp16) — the packer says so."""
    w = torch.randn(15, 3, 3, 3)
    with pytest.raises(NotImplementedError):
        E.PackedGroupFilter(w.to(dev), 5, torch.float16)


# ---- 3x3 convolutions on the antiphase GEMM kernel (gemm_pp.hip CONV mode: candidate 7 = 256 x 256 tiles,
# candidate 9 = 128 x 256), forced through TLXMI_TILE and checked against the oracle: every tap-mask edge (image
# borders, image-to-image boundaries inside a tile, stride 2, odd extents), 1 / 2 / 4 / 8 K tiles per tap, channel
# tails, row tails, residual + activation epilogues.
PPCONV_CASES = [
    # (N, Cin, Cout, k, stride, pad, H, W, res, act)
    (2, 256, 256, 3, 1, 1, 14, 14, False, 1),      # resnet.py:111-121 conv2 of layer3
    (3, 512, 512, 3, 1, 1, 7, 7, False, 1),        # layer4: several images inside one 256-row tile
    (1, 256, 256, 3, 2, 1, 28, 28, False, 1),      # stride-2 entry of layer3
    (1, 512, 512, 3, 2, 1, 15, 13, True, 1),       # odd extents, stride 2, residual
    (2, 64, 256, 3, 1, 1, 12, 20, False, 3),       # one K tile per tap (fp16), leaky
    (1, 128, 264, 3, 1, 0, 19, 19, True, 0),       # no padding, channel tail (264 = 256 + 8)
    (1, 128, 512, (1, 3), 1, (0, 1), 9, 33, False, 2),   # 1 x 3 filter
    (5, 128, 256, 3, 1, 1, 10, 10, False, 1),      # 500 rows: 2 tiles of 256, the second ragged
    (2, 256, 512, 1, 2, 0, 28, 28, False, 0),      # strided 1x1 projection shortcut (resnet.py:246-261): one tap, rows at stride 2
    (1, 512, 1024, 1, 2, 0, 15, 13, False, 1),     # odd extents
    (3, 128, 256, 1, 2, 0, 10, 10, True, 0),
]


@pytest.mark.parametrize("tile", ["7", "9", "10"], ids=["pp256", "pp128x256", "pp256x128"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
@pytest.mark.parametrize("cfg", PPCONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv3x3_on_the_antiphase_gemm(dev, dtype, cfg, tile):
    N, Cin, Cout, k, stride, pad, H, W, res, act = cfg
    with tuning(TLXMI_TILE=tile):
        run_case(dev, dtype, N, Cin, Cout, k, stride, pad, H, W, act=act, act_param=0.1, with_res=res, seed=23)


# 128 output channels: only the 256 x 128 tile (candidate 10) takes these (resnet.py:111-121, 28 x 28 stage)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
@pytest.mark.parametrize("cfg", [(2, 128, 128, 3, 1, 1, 28, 28, False, 1), (1, 128, 128, 3, 2, 1, 56, 56, False, 1),
                                 (3, 64, 136, 3, 1, 1, 11, 17, True, 3), (1, 256, 128, 3, 1, 1, 16, 16, True, 0)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv3x3_with_128_output_channels_on_the_antiphase_gemm(dev, dtype, cfg):
    N, Cin, Cout, k, stride, pad, H, W, res, act = cfg
    with tuning(TLXMI_TILE="10"):
        run_case(dev, dtype, N, Cin, Cout, k, stride, pad, H, W, act=act, act_param=0.1, with_res=res, seed=29)


def test_linear_on_the_256x128_antiphase_tile(dev):
    """The 1x1 / Linear path of the same tile shape (reachable through TLXMI_TILE=10 only)."""
    with tuning(TLXMI_TILE="10"):
        run_case(dev, torch.float16, 2, 512, 128, 1, 1, 0, 28, 28, act=1, with_res=False, seed=31)
        run_case(dev, torch.float32, 1, 256, 392, 1, 1, 0, 19, 19, act=0, with_res=True, seed=32)


def test_conv3x3_image_axis_tail_split(dev):
    """270 tiles of 256 x 128 on 256 CUs: the dispatcher keeps 83 images on the antiphase kernel and runs the last 5 as
    a convolution of their own on small tiles; the seam (an image boundary) must be invisible.  Residual + ReLU."""
    run_case(dev, torch.float16, 88, 128, 128, 3, 1, 1, 28, 28, act=1, with_res=True, seed=37)


# ---- padding='SAME' at stride 2 (efficientnet.py:92-125): TensorFlow's rule puts the odd unit of padding at the bottom /
# right, i.e. the last windows run past the edge (one-sided end padding of tlxmi_conv2d / tlxmi_dwconv2d)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
@pytest.mark.parametrize("cfg", [(2, 8, 32, 3, 2, 1, 24, 24), (1, 16, 24, 3, 2, 1, 17, 31), (1, 96, 96, 3, 2, 96, 28, 28),
                                 (2, 144, 144, 5, 2, 144, 14, 14), (1, 40, 40, 5, 2, 40, 15, 9), (1, 64, 64, 3, 1, 64, 12, 12),
                                 (1, 32, 64, 2, 1, 1, 9, 9), (1, 24, 24, 4, 2, 1, 10, 10)],
                         ids=lambda c: "x".join(map(str, c)))
def test_same_padding_one_sided(dev, dtype, cfg):
    import tlxcv_amd
    from tlxcv_amd.tlx import nn
    N, Cin, Cout, k, stride, groups, H, W = cfg
    tlxcv_amd.set_precision("fp32" if dtype == torch.float32 else "fp16")
    try:
        rng = np.random.default_rng(43)
        conv = nn.GroupConv2d(Cout, (k, k), (stride, stride), groups, None, "SAME", in_channels=Cin, data_format="channels_first")
        w, bias = rnd(rng, (Cout, Cin // groups, k, k), (2.0 / (Cin // groups * k * k)) ** 0.5), rnd(rng, (Cout,), 0.1)
        x = rnd(rng, (N, Cin, H, W))
        if dtype == torch.float16:
            w, x = q16(w), q16(x)
        conv.load_dict({"filters": w, "biases": bias})
        conv = conv.to(dev).set_eval()
        want = torch.nn.functional.conv2d(OF.same_pad(x, k, stride), w, bias, stride, 0, 1, groups)
        got = conv.run_nhwc(nchw_to_engine(x, dtype, dev))
        torch.cuda.synchronize()
        assert got.shape[1:3] == (-(-H // stride), -(-W // stride))
        torch.testing.assert_close(engine_to_nchw(got), want, **tol(dtype))
    finally:
        tlxcv_amd.set_precision("fp16")


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
@pytest.mark.parametrize("k,H,W", [(3, 64, 64), (3, 30, 46), (5, 32, 32)])
def test_space_to_depth_stem_with_same_padding(dev, dtype, k, H, W):
    """The 'SAME' stride-2 RGB stem (efficientnet.py:354-363) on the 2 x 2 space-to-depth image: leading padding from
    TensorFlow's rule, the odd unit as windows past the bottom / right edge."""
    import tlxcv_amd
    from tlxcv_amd.tlx import nn
    tlxcv_amd.set_precision("fp32" if dtype == torch.float32 else "fp16")
    try:
        rng = np.random.default_rng(59)
        conv = nn.GroupConv2d(32, (k, k), (2, 2), 1, None, "SAME", in_channels=3, data_format="channels_first")
        w, bias = rnd(rng, (32, 3, k, k), 0.2), rnd(rng, (32,), 0.1)
        x = rnd(rng, (2, 3, H, W))
        if dtype == torch.float16:
            w, x = q16(w), q16(x)
        conv.load_dict({"filters": w, "biases": bias})
        conv = conv.to(dev).set_eval()
        want = torch.nn.functional.conv2d(OF.same_pad(x, k, 2), w, bias, 2)
        got = conv.run_stem(x.to(dev), 2)
        torch.cuda.synchronize()
        torch.testing.assert_close(engine_to_nchw(got), want, **tol(dtype))
    finally:
        tlxcv_amd.set_precision("fp16")


@pytest.mark.parametrize("N,act", [(1, E.ACT_RELU), (3, E.ACT_RELU), (5, E.ACT_LEAKY), (9, E.ACT_NONE)])
def test_stem_with_the_max_pool_in_its_epilogue(dev, N, act):
    """resnet.py:287-290 conv1 -> bn1 -> relu -> maxpool(3, 2, 1) as ONE launch (TLXMI_EPI_MAXPOOL_3S2P1): must equal the
    two-launch path bit for bit (the conv result is rounded to fp16 once in both, a maximum is exact), and the oracle.
    The batch sizes make workgroup ranges start at an image's first tile, inside an image (recomputed warm-up tile) and
    span image seams; activations without a lower bound of 0 check that the pool's padding is 'skip', not zero."""
    from tlxcv_amd.tlx import nn
    rng = np.random.default_rng(N)
    conv = nn.GroupConv2d(in_channels=3, out_channels=64, kernel_size=7, stride=2, padding=3, b_init=None,
                          data_format="channels_first")
    bn = nn.BatchNorm2d(num_features=64, data_format="channels_first")
    with torch.no_grad():
        conv.filters.copy_(q16(rnd(rng, (64, 3, 7, 7), (2.0 / 147) ** 0.5)))
        bn.gamma.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, 64).astype(np.float32)))
        bn.beta.copy_(rnd(rng, (64,), 0.3) - 0.5)           # mostly negative maps: a zero-padded pool would differ
        bn.moving_mean.copy_(rnd(rng, (64,), 0.1))
        bn.moving_var.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, 64).astype(np.float32)))
    conv, bn = conv.to(dev).set_eval(), bn.to(dev).set_eval()
    pool = nn.MaxPool2d(3, 2, 1, data_format="channels_first")
    x = q16(rnd(rng, (N, 3, 224, 224)))
    fused = conv.run_stem(x.to(dev), 2, bn, act, 0.1, maxpool=pool)
    two = pool.run_nhwc(conv.run_stem(x.to(dev), 2, bn, act, 0.1))
    torch.cuda.synchronize()
    assert fused.shape == two.shape == (N, 56, 56, 64)
    assert torch.equal(fused, two)
    scale, shift = bn.folded(None)
    want = torch.nn.functional.max_pool2d(
        OF.conv_bn_act(x, conv.filters.cpu(), scale.cpu(), shift.cpu(), None, act, 0.1, 2, 3), 3, 2, 1)
    torch.testing.assert_close(engine_to_nchw(fused), want, **tol(torch.float16))
    # geometries without the fused kernel fall back to two launches inside run_stem (nothing else changes for the caller)
    small = conv.run_stem(x[:1, :, :64, :64].contiguous().to(dev), 2, bn, act, 0.1, maxpool=pool)
    assert small.shape == (1, 16, 16, 64)


# ---- few output pixels, long K: K slices side by side + reduction (tlxmi_conv2d_splitk; ResNet's 7 x 7 stage at small batch)
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.float32], ids=["fp16", "fp32"])
@pytest.mark.parametrize("cfg", [(4, 7, 512, 512, 3, 1, 1, True), (32, 7, 512, 512, 3, 1, 1, False), (2, 14, 512, 256, 3, 2, 1, True),
                                 (3, 9, 1024, 384, 3, 1, 1, False)], ids=lambda c: "x".join(map(str, c)))
def test_conv2d_splitk_matches_the_oracle_and_the_single_launch(dev, dtype, cfg):
    from tlxcv_amd import _lib
    import ctypes as C
    N, H, Cin, Cout, R, s, p, with_res = cfg
    rng = np.random.default_rng(17)
    x = torch.from_numpy(rng.standard_normal((N, H, H, Cin)).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((Cout, Cin, R, R)) * (2.0 / (Cin * R * R)) ** 0.5).astype(np.float32))
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, Cout).astype(np.float32))
    sh = torch.from_numpy(rng.standard_normal(Cout).astype(np.float32))
    Ho = (H + 2 * p - R) // s + 1
    r = torch.from_numpy(rng.standard_normal((N, Ho, Ho, Cout)).astype(np.float32)) if with_res else None
    if dtype == torch.float16:
        x, w = x.half().float(), w.half().float()
        r = r.half().float() if r is not None else None
    want = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, None, s, p).permute(0, 2, 3, 1) * sc + sh
    if r is not None:
        want = want + r
    want = torch.relu(want)
    pk = E.PackedFilter(w.to(dev), dtype)
    xd, rd = x.to(dtype).to(dev), (r.to(dtype).to(dev) if r is not None else None)
    scd, shd = sc.to(dev), sh.to(dev)            # (kept alive: the C-ABI takes raw pointers)
    out = torch.empty((N, Ho, Ho, Cout), dtype=dtype, device=dev)
    d = _lib.ConvDesc(dtype=E.dt_code(dtype), N=N, H=H, W=H, C=pk.Cin_pad, Cout=Cout, R=R, S=R, stride_h=s, stride_w=s, pad_h=p, pad_w=p,
                      dil_h=1, dil_w=1, Ho=Ho, Wo=Ho, x_ld=Cin, y_ld=Cout, res_ld=Cout if r is not None else 0, y_nstride=0, res_nstride=0,
                      act=E.ACT_RELU, act_param=0.0, flags=0)
    lib = _lib.load()
    for splits in (2, 4):
        assert lib.tlxmi_conv2d_splitk_supported(C.byref(d), splits) == 1
        part = torch.full((splits, N * Ho * Ho, Cout), float("nan"), dtype=torch.float32, device=dev)
        out.fill_(float("nan"))
        _lib.call("tlxmi_conv2d_splitk", C.byref(d), splits, E._p(xd), E._p(pk.buf), E._p(part), E._p(scd), E._p(shd),
                  E._p(rd), E._p(out), E._stream())
        torch.cuda.synchronize()
        got = out.float().cpu()
        tol = 1e-4 if dtype == torch.float32 else 4e-3
        assert float((got - want).abs().max()) <= tol * max(1.0, float(want.abs().max())), splits
    E.set_option("conv_splitk", False)
    try:
        one = E.conv2d(xd, pk, s, p, 1, scd, shd, rd, E.ACT_RELU).float().cpu()
    finally:
        E.set_option("conv_splitk", True)
    assert float((got - one).abs().max()) <= (1e-4 if dtype == torch.float32 else 4e-3) * max(1.0, float(want.abs().max()))


@pytest.mark.gpu
def test_conv2d_splitk_refuses_what_it_cannot_slice(dev):
    from tlxcv_amd import _lib
    import ctypes as C
    lib = _lib.load()

    def desc(**kw):
        base = dict(dtype=E.dt_code(torch.float16), N=4, H=7, W=7, C=512, Cout=512, R=3, S=3, stride_h=1, stride_w=1, pad_h=1, pad_w=1,
                    dil_h=1, dil_w=1, Ho=7, Wo=7, x_ld=512, y_ld=512, res_ld=0, y_nstride=0, res_nstride=0, act=0, act_param=0.0, flags=0)
        base.update(kw)
        return _lib.ConvDesc(**base)
    assert lib.tlxmi_conv2d_splitk_supported(C.byref(desc()), 2) == 1
    assert lib.tlxmi_conv2d_splitk_supported(C.byref(desc()), 1) == 0                    # not a split
    assert lib.tlxmi_conv2d_splitk_supported(C.byref(desc()), 32) == 0                   # fewer than 4 K tiles per slice
    assert lib.tlxmi_conv2d_splitk_supported(C.byref(desc(C=96, x_ld=96)), 2) == 0       # a tap is not whole K tiles
    assert lib.tlxmi_conv2d_splitk_supported(C.byref(desc(Cout=64, y_ld=64)), 2) == 0    # too few output channels for the GEMM tile
    assert lib.tlxmi_conv2d_splitk_supported(C.byref(desc(dil_h=2, dil_w=2)), 2) == 0
    assert lib.tlxmi_conv2d_splitk_supported(C.byref(desc(R=1, S=1, pad_h=0, pad_w=0)), 2) == 0   # a plain 1x1: tlxmi_linear_splitk's job
    d = desc(Cout=64, y_ld=64)
    x = torch.zeros((4, 7, 7, 512), dtype=torch.float16, device=dev)
    with pytest.raises(RuntimeError, match="not supported"):
        _lib.call("tlxmi_conv2d_splitk", C.byref(d), 2, E._p(x), E._p(x), E._p(x), None, None, None, E._p(x), E._stream())


# ---- maximum sizes: the kernels address their tensors with 32-bit buffer offsets (2 GiB per tensor).  A descriptor beyond
# that is refused by the entry point with a message — before any launch, so the buffers handed over here can be tiny
def test_tensors_beyond_the_32_bit_offsets_are_refused_not_launched(dev):
    from tlxcv_amd import _lib
    tiny = torch.zeros(64, device=dev, dtype=torch.float16)
    f32 = torch.zeros(64, device=dev, dtype=torch.float32)
    p = E._p
    pk = E.PackedFilter(torch.zeros((64, 64, 1, 1), device=dev), torch.float16)
    d = _lib.ConvDesc(dtype=E.dt_code(torch.float16), N=4096, H=224, W=224, C=64, Cout=64, R=1, S=1, stride_h=1, stride_w=1, pad_h=0,
                      pad_w=0, dil_h=1, dil_w=1, Ho=224, Wo=224, x_ld=64, y_ld=64, res_ld=0, y_nstride=0, res_nstride=0, act=0,
                      act_param=0.0, flags=0)                       # 26 GB of input
    with pytest.raises(RuntimeError, match="2 GiB"):
        _lib.call("tlxmi_conv2d", ctypes.byref(d), p(tiny), p(pk.buf), None, None, None, p(tiny), E._stream())
    assert _lib.load().tlxmi_group_conv2d_small_supported(ctypes.byref(d), 16) == 0
    sd = _lib.SeamDesc(dtype=E.dt_code(torch.float16), rows=1 << 23, K1=64, N1=256, N2=64, t2_ld=64, skip_ld=256, y_ld=256, t1_ld=64,
                       act=E.ACT_RELU)                             # y: 2^23 rows x 512 bytes = 4 GiB
    with pytest.raises(RuntimeError, match="2 GiB"):
        _lib.call("tlxmi_bottleneck_seam", ctypes.byref(sd), p(tiny), p(pk.buf), p(f32), p(f32), p(tiny), p(tiny), p(pk.buf), p(f32),
                  p(f32), p(tiny), E._stream())
    torch.cuda.synchronize()
