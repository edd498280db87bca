"""HBM-bound kernels and attention vs the CPU oracle (torch fp32 on the same seeded inputs)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import functional as OF
from tlxcv_amd import engine as E
from util import rnd, q16, nchw_to_engine, engine_to_nchw, tol

pytestmark = pytest.mark.gpu
DT = pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])


def prep(t, dtype):
    return q16(t) if dtype == torch.float16 else t


@DT
@pytest.mark.parametrize("shape", [(2, 3, 17, 23), (1, 64, 8, 8), (3, 20, 5, 7)])
def test_layout_roundtrip_and_padding(dev, dtype, shape):
    rng = np.random.default_rng(0)
    x = prep(rnd(rng, shape), dtype)
    y = E.nchw_to_nhwc(x.to(dev), dtype)
    v = E.vec(dtype)
    cp = (shape[1] + v - 1) // v * v
    assert y.shape == (shape[0], shape[2], shape[3], cp)
    got = y.float().cpu()
    assert torch.equal(got[..., :shape[1]], x.permute(0, 2, 3, 1))
    assert (got[..., shape[1]:] == 0).all()
    back = E.nhwc_to_nchw(y, shape[1], torch.float32)
    assert torch.equal(back.cpu(), x)


@DT
def test_maxpool_resnet_stem_and_borders(dev, dtype):
    rng = np.random.default_rng(1)
    for (N, C, H, W, k, s, p) in [(2, 64, 112, 112, 3, 2, 1), (1, 8, 7, 9, 3, 2, 1), (1, 16, 6, 6, 2, 2, 0)]:
        x = prep(rnd(rng, (N, C, H, W)) - 3.0, dtype)       # all-negative rows: -inf padding must not win as 0
        got = E.maxpool2d(nchw_to_engine(x, dtype, dev), k, s, p)
        torch.testing.assert_close(engine_to_nchw(got), F.max_pool2d(x, k, s, p), atol=0, rtol=0)


@DT
def test_global_avgpool(dev, dtype):
    rng = np.random.default_rng(2)
    x = prep(rnd(rng, (3, 2048, 7, 7)), dtype)
    got = E.global_avgpool(nchw_to_engine(x, dtype, dev))
    torch.testing.assert_close(got.float().cpu(), x.mean((2, 3)), **tol(dtype))


@DT
@pytest.mark.parametrize("act", [0, 1, 3, 6])
def test_affine_act(dev, dtype, act):
    rng = np.random.default_rng(3)
    x, r = prep(rnd(rng, (37, 72)), dtype), prep(rnd(rng, (37, 72)), dtype)
    sc, sh = rnd(rng, (72,)), rnd(rng, (72,))
    from oracle.functional import ACTS
    want = ACTS[act](x * sc + sh + r, 0.2)
    got = E.affine_act(x.to(dtype).to(dev), sc.to(dev), sh.to(dev), r.to(dtype).to(dev), act, 0.2)
    torch.testing.assert_close(got.float().cpu(), want, **tol(dtype))


@DT
@pytest.mark.parametrize("C,eps", [(768, 1e-6), (128, 1e-5), (4096, 1e-5), (24, 1e-5)])
def test_layernorm(dev, dtype, C, eps):
    rng = np.random.default_rng(4)
    x = prep(rnd(rng, (2, 19, C), 2.0) + 0.5, dtype)
    g, b = rnd(rng, (C,)) * 0.1 + 1, rnd(rng, (C,), 0.1)
    got = E.layernorm(x.to(dtype).to(dev), g.to(dev), b.to(dev), eps)
    torch.testing.assert_close(got.float().cpu(), F.layer_norm(x, (C,), g, b, eps), **tol(dtype))


@DT
@pytest.mark.parametrize("k,s,p,act", [(3, 1, 1, 1), (3, 2, 1, 2), (5, 1, 2, 4), (5, 2, 2, 1)])
def test_depthwise_conv(dev, dtype, k, s, p, act):
    rng = np.random.default_rng(5)
    C = 40
    x = prep(rnd(rng, (2, C, 15, 13)), dtype)
    w = prep(rnd(rng, (C, 1, k, k), 0.3), dtype)
    sc, sh = torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32)), rnd(rng, (C,), 0.1)
    from oracle.functional import conv_bn_act
    want = conv_bn_act(x, w, sc, sh, None, act, 0.0, s, p, 1, C)
    w_rsc = w[:, 0].permute(1, 2, 0).contiguous().to(dtype).to(dev)
    got = E.dwconv2d(nchw_to_engine(x, dtype, dev), w_rsc, s, p, 1, sc.to(dev), sh.to(dev), act)
    torch.testing.assert_close(engine_to_nchw(got), want, **tol(dtype))


def _ref_attention(qkv, heads, scale, bias=None, mask=None):
    """vision_transformer.py:112-120 / swin_transformer.py:194-224 on torch CPU."""
    B, N, C3 = qkv.shape
    hd = C3 // 3 // heads
    t = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = t[0], t[1], t[2]
    attn = (q @ k.transpose(-1, -2)) * scale
    if bias is not None:
        attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.reshape(B // nW, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)).reshape(-1, heads, N, N)
    attn = torch.softmax(attn, -1)
    return (attn @ v).permute(0, 2, 1, 3).reshape(B, N, heads * hd)


@DT
@pytest.mark.parametrize("B,N,heads,hd,swin", [(3, 197, 12, 64, False), (2, 50, 3, 64, False), (8, 49, 4, 32, True),
                                                (4, 49, 32, 32, True), (2, 16, 2, 32, False), (1, 256, 1, 64, False),
                                                (2, 577, 3, 64, False), (4, 300, 2, 96, True), (1, 1025, 1, 128, False),
                                                (3, 197, 8, 96, False), (4, 49, 2, 96, True)])
def test_attention(dev, dtype, B, N, heads, hd, swin):
    rng = np.random.default_rng(6)
    qkv = prep(rnd(rng, (B, N, 3 * heads * hd)), dtype)
    bias = rnd(rng, (heads, N, N), 0.5) if swin else None
    mask = None
    if swin:
        nW = 4
        ids = torch.from_numpy(rng.integers(0, 3, (nW, N)))
        mask = (ids.unsqueeze(1) != ids.unsqueeze(2)).float() * -100.0        # swin_transformer.py:303-305
    scale = hd ** -0.5
    want = _ref_attention(qkv, heads, scale, bias, mask)
    got = E.attention(qkv.to(dtype).to(dev), heads, scale, bias.to(dev) if swin else None,
                      mask.to(dev) if swin else None)
    t = tol(dtype) if dtype == torch.float32 else dict(atol=4e-3, rtol=4e-3)
    torch.testing.assert_close(got.float().cpu(), want, **t)


@pytest.mark.parametrize("N", [65, 96, 100, 128, 129, 150, 197, 200, 224, 225, 256])
def test_vit_attention_with_k_v_staged_by_lds_dma(dev, N):
    """attn_dma_kernel (head dim 64, no bias / mask, 65 .. 256 tokens: every key-tile count and both padding-mask variants):
    against the fp32 reference on the fp16-rounded inputs, against the register-staged kernel it replaces (TLXMI_ATTN_DMA=0,
    tuning flavour) bit for bit — same arithmetic, only the staging differs — and reproducible."""
    from tlxcv_amd._lib import tuning
    rng = np.random.default_rng(N)
    B, heads, hd = 5, 3, 64
    qkv = prep(rnd(rng, (B, N, 3 * heads * hd)), torch.float16)
    want = _ref_attention(qkv, heads, hd ** -0.5, None, None)
    x = qkv.half().to(dev)
    got = E.attention(x, heads, hd ** -0.5)
    torch.testing.assert_close(got.float().cpu(), want, atol=4e-3, rtol=4e-3)
    assert torch.equal(E.attention(x, heads, hd ** -0.5), got)
    with tuning(TLXMI_ATTN_DMA="0"):
        old = E.attention(x, heads, hd ** -0.5)
    with tuning(TLXMI_ATTN_DMA="1"):
        new = E.attention(x, heads, hd ** -0.5)
    torch.cuda.synchronize()
    assert torch.equal(old, new) and torch.equal(new, got)


@pytest.mark.parametrize("B,N,heads,hd,masked", [(8, 49, 4, 32, True), (4, 49, 16, 32, False), (6, 130, 2, 64, True), (2, 197, 3, 96, False)])
def test_attention_with_presummed_table(dev, B, N, heads, hd, masked):
    """tlxmi_attention_comb: bias + mask summed and padded by the caller (swin_transformer.py:205-220)."""
    rng = np.random.default_rng(16)
    qkv = q16(rnd(rng, (B, N, 3 * heads * hd)))
    bias = rnd(rng, (heads, N, N), 0.5)
    mask = None
    if masked:
        nW = 2
        ids = torch.from_numpy(rng.integers(0, 3, (nW, N)))
        mask = (ids.unsqueeze(1) != ids.unsqueeze(2)).float() * -100.0
    scale = hd ** -0.5
    want = _ref_attention(qkv, heads, scale, bias, mask)
    tab = E.attention_table(bias.to(dev), mask.to(dev) if masked else None, N)
    got = E.attention_comb(qkv.half().to(dev), heads, scale, tab, mask.shape[0] if masked else 0)
    torch.testing.assert_close(got.float().cpu(), want, atol=4e-3, rtol=4e-3)


@pytest.mark.parametrize("B,N,heads,hd", [(2, 577, 3, 64), (1, 257, 2, 64), (3, 300, 4, 32), (1, 1025, 2, 64), (2, 640, 1, 32)])
def test_long_sequence_attention_on_the_chunked_online_softmax_kernel(dev, B, N, heads, hd):
    """More than 256 tokens (ViT at 384 x 384: 577, vision_transformer.py:358-): keys in chunks of 128 through LDS, running
    maximum / sum per query tile.  577 = 4 chunks + 65 keys, 257 = one key in the last chunk, 640 = whole chunks only."""
    rng = np.random.default_rng(60 + N)
    qkv = rnd(rng, (B, N, 3 * heads * hd))
    # a dominant key far into the sequence for some queries: the running maximum jumps in a LATER chunk (the rescale path)
    qkv[0, 3, :hd] = 4.0
    qkv[0, N - 2, heads * hd:heads * hd + hd] = 5.0
    qkv[0, 200, :hd] = -3.0
    qkv = q16(qkv)
    want = _ref_attention(qkv, heads, hd ** -0.5)
    got = E.attention(qkv.half().to(dev), heads, hd ** -0.5)
    assert torch.isfinite(got).all()
    torch.testing.assert_close(got.float().cpu(), want, atol=4e-3, rtol=4e-3)
    assert torch.equal(E.attention(qkv.half().to(dev), heads, hd ** -0.5), got)


def test_attention_softmax_is_stable_for_large_scores(dev):
    """Forces the max-subtraction path: one key dominates with a score ~ +80 (exp overflows in fp16)."""
    rng = np.random.default_rng(7)
    B, N, heads, hd = 1, 64, 1, 64
    qkv = rnd(rng, (B, N, 3 * hd), 0.1)
    qkv[0, 5, :hd] = 8.0          # q row 5
    qkv[0, 9, hd:2 * hd] = 10.0   # k row 9  -> score 8*10*64/8 = 640*... scaled
    qkv = q16(qkv)
    want = _ref_attention(qkv, heads, hd ** -0.5)
    got = E.attention(qkv.half().to(dev), heads, hd ** -0.5)
    assert torch.isfinite(got).all()
    torch.testing.assert_close(got.float().cpu(), want, atol=4e-3, rtol=4e-3)


@DT
@pytest.mark.parametrize("shift", [0, 3])
def test_window_partition_reverse(dev, dtype, shift):
    rng = np.random.default_rng(8)
    B, H, W, C, ws = 2, 14, 14, 32, 7
    x = prep(rnd(rng, (B, H, W, C)), dtype)
    xs = torch.roll(x, (-shift, -shift), (1, 2)) if shift else x                       # swin :317-319
    want = xs.reshape(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)  # :85-99
    win = E.window_partition(x.to(dtype).to(dev), ws, shift)
    assert torch.equal(win.float().cpu(), want)
    back = E.window_reverse(win, B, H, W, ws, shift)
    assert torch.equal(back.float().cpu(), x)                                          # reverse o partition = id
    r = prep(rnd(rng, (B, H, W, C)), dtype)
    fused = E.window_reverse(win, B, H, W, ws, shift, res=r.to(dtype).to(dev))
    torch.testing.assert_close(fused.float().cpu(), x + r, **tol(dtype))


@DT
@pytest.mark.parametrize("shift", [0, 3])
@pytest.mark.parametrize("geom", [(2, 14, 14, 128, 7), (1, 28, 56, 512, 7), (3, 7, 7, 1024, 7)], ids=lambda g: "x".join(map(str, g)))
def test_layernorm_fused_with_window_plumbing(dev, dtype, shift, geom):
    """norm1 + roll + window_partition, and window_reverse + roll back + residual + norm2 (swin :315-335) in one
    pass each, against the separate steps on the CPU."""
    import torch.nn.functional as F
    B, H, W, C, ws = geom
    rng = np.random.default_rng(12)
    x = prep(rnd(rng, (B, H, W, C)) + 0.3, dtype)
    g = torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32))
    b = rnd(rng, (C,), 0.2)
    part = lambda t: (torch.roll(t, (-shift, -shift), (1, 2)) if shift else t).reshape(
        B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
    want_win = part(F.layer_norm(x, (C,), g, b, 1e-5))
    got_win = E.layernorm_window_partition(x.to(dtype).to(dev), g.to(dev), b.to(dev), 1e-5, ws, shift)
    torch.testing.assert_close(got_win.float().cpu(), want_win, **(dict(atol=1e-4, rtol=1e-4) if dtype == torch.float32 else dict(atol=4e-3, rtol=4e-3)))
    # reverse: win holds partition(a); sum = x + a, y = LN(sum)
    a = prep(rnd(rng, (B, H, W, C)), dtype)
    win = part(a).contiguous()
    s_got, y_got = E.window_reverse_layernorm(win.to(dtype).to(dev), x.to(dtype).to(dev), g.to(dev), b.to(dev), 1e-5, ws, shift)
    s_want = prep(x + a, dtype)                      # the sum is stored in the engine dtype before norm2 reads it
    torch.testing.assert_close(s_got.float().cpu(), s_want, **tol(dtype))
    y_want = F.layer_norm(s_got.float().cpu(), (C,), g, b, 1e-5)
    torch.testing.assert_close(y_got.float().cpu(), y_want, **(dict(atol=1e-4, rtol=1e-4) if dtype == torch.float32 else dict(atol=4e-3, rtol=4e-3)))


@DT
def test_patch_merge_gather(dev, dtype):
    rng = np.random.default_rng(9)
    x = prep(rnd(rng, (2, 8, 6, 16)), dtype)
    want = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)   # swin :381-386
    got = E.patch_merge_gather(x.to(dtype).to(dev))
    assert torch.equal(got.float().cpu(), want)


@DT
@pytest.mark.parametrize("shape", [(2, 8, 6, 16), (3, 14, 14, 128), (5, 28, 28, 256), (2, 6, 10, 24), (1, 4, 4, 512)], ids=lambda s: "x".join(map(str, s)))
def test_patch_merge_layernorm(dev, dtype, shape):
    """tlxmi_patch_merge_layernorm (swin_transformer.py:381-388): bit for bit the two launches it replaces (gather, then LayerNorm
    over 4C), and the torch reference on the same rounded input; widths of every LayerNorm lane layout (64 .. 2048 channels)."""
    rng = np.random.default_rng(11)
    x = prep(rnd(rng, shape), dtype)
    B, H, W, Cc = shape
    g, b = rnd(rng, (4 * Cc,), 0.5) + 1.0, rnd(rng, (4 * Cc,), 0.5)
    xd = x.to(dtype).to(dev)
    got = E.patch_merge_layernorm(xd, g.to(dev), b.to(dev), 1e-5)
    two = E.layernorm(E.patch_merge_gather(xd).view(B, (H // 2) * (W // 2), 4 * Cc), g.to(dev), b.to(dev), 1e-5)
    assert got.shape == (B, (H // 2) * (W // 2), 4 * Cc) and torch.equal(got, two)
    cat = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1).reshape(B, -1, 4 * Cc)
    want = F.layer_norm(cat, (4 * Cc,), g, b, 1e-5)
    torch.testing.assert_close(got.float().cpu(), want, **(dict(atol=1e-4, rtol=1e-4) if dtype == torch.float32 else dict(atol=4e-3, rtol=4e-3)))


@DT
@pytest.mark.parametrize("shape,ps,lead", [((2, 3, 32, 48), 16, 1), ((3, 3, 64, 64), 32, 0), ((1, 4, 16, 24), 8, 2), ((5, 3, 224, 224), 16, 1)],
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else str(v))
def test_patchify(dev, dtype, shape, ps, lead):
    """tlxmi_patchify: patch rows in the flattened-filter order, zero rows in front of each image — exact (a copy + one rounding);
    and conv2d(x, w, stride = ps) == rows @ w.reshape(Cout, -1).T, the identity vision_transformer.py's PatchEmbed path relies on."""
    rng = np.random.default_rng(12)
    x = rnd(rng, shape)
    N, Cc, H, W = shape
    got = E.patchify(x.to(dev), ps, lead, dtype)
    Hp, Wp = H // ps, W // ps
    want = x.reshape(N, Cc, Hp, ps, Wp, ps).permute(0, 2, 4, 1, 3, 5).reshape(N, Hp * Wp, Cc * ps * ps).to(dtype)
    assert got.shape == (N, lead + Hp * Wp, Cc * ps * ps)
    assert torch.equal(got[:, lead:].cpu(), want) and not got[:, :lead].any()
    w = rnd(rng, (8, Cc, ps, ps))
    torch.testing.assert_close(F.conv2d(want.float().reshape(N, Hp, Wp, Cc, ps, ps).permute(0, 3, 1, 4, 2, 5).reshape(N, Cc, H, W), w, stride=ps)
                               .permute(0, 2, 3, 1).reshape(N, Hp * Wp, 8), want.float() @ w.reshape(8, -1).t(), atol=2e-3, rtol=2e-3)


@pytest.mark.parametrize("D", [96, 128, 192, 256])
@pytest.mark.parametrize("shape,norm,xdt", [((2, 3, 32, 48), True, torch.float32), ((3, 3, 224, 224), True, torch.float32),
                                            ((1, 3, 8, 12), False, torch.float32), ((2, 3, 56, 40), True, torch.float16)],
                         ids=["2x32x48", "3x224x224", "1x8x12-nonorm", "2x56x40-fp16in"])
def test_swin_patch_embedding_in_one_pass(dev, fp16_mode, D, shape, norm, xdt):
    """tlxmi_patch_embed4 (swin_transformer.py:471-505: conv 4 x 4 / 4 + bias -> tokens -> LayerNorm) against torch in fp32 on the
    fp16-rounded image and filter; a token count that is not a multiple of the 16-token wave tile; reproducible."""
    rng = np.random.default_rng(13)
    x = rnd(rng, shape).to(xdt)
    w, b = rnd(rng, (D, 3, 4, 4), 0.3), rnd(rng, (D,), 0.5)
    g, be = rnd(rng, (D,), 0.5) + 1.0, rnd(rng, (D,), 0.5)
    w64 = E.patch_embed4_filter(w.to(dev))
    assert w64.shape == (D, 64) and not w64[:, 48:].any()
    got = E.patch_embed4(x.to(dev), w64, b.to(dev), g.to(dev) if norm else None, be.to(dev) if norm else None, 1e-5)
    N, _, H, W = shape
    assert got.shape == (N, (H // 4) * (W // 4), D) and got.dtype == torch.float16
    want = F.conv2d(x.half().float(), w.half().float(), b, stride=4).flatten(2).transpose(1, 2)
    if norm:
        want = F.layer_norm(want, (D,), g, be, 1e-5)
    torch.testing.assert_close(got.float().cpu(), want, atol=6e-3, rtol=6e-3)
    assert torch.equal(E.patch_embed4(x.to(dev), w64, b.to(dev), g.to(dev) if norm else None, be.to(dev) if norm else None, 1e-5), got)


@DT
def test_upsample_concat(dev, dtype):
    rng = np.random.default_rng(10)
    a, b = prep(rnd(rng, (1, 16, 5, 4)), dtype), prep(rnd(rng, (1, 24, 10, 8)), dtype)
    want = torch.cat([F.interpolate(a, scale_factor=2, mode="nearest"), b], 1)                         # yolov3.py:250-256
    out = torch.empty((1, 10, 8, 40), dtype=dtype, device=dev)
    E.upsample2x_into(nchw_to_engine(a, dtype, dev), out, 0)
    E.copy_channels_into(nchw_to_engine(b, dtype, dev), out, 16)
    assert torch.equal(engine_to_nchw(out), want)


def test_argmax_ties_nan_and_tail(dev):
    x = torch.tensor([[1.0, 5.0, 5.0, 2.0], [3.0, float("nan"), 9.0, 1.0], [-2.0, -1.0, -3.0, -1.0]])
    got = E.argmax_lastdim(x.to(dev)).cpu()
    assert got.tolist() == torch.argmax(x, -1).tolist() == [1, 1, 1]
    g = torch.Generator().manual_seed(0)
    big = torch.randn((257, 1000), generator=g)
    assert torch.equal(E.argmax_lastdim(big.half().to(dev)).cpu(), torch.argmax(big.half().float(), -1))


@pytest.mark.parametrize("shape", [(3, 3, 30, 46), (1, 3, 224, 224), (2, 3, 6, 2), (2, 4, 8, 8)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("offset", [0, 1], ids=["aligned", "odd_element_offset"])
@pytest.mark.parametrize("src", [torch.float32, torch.float16], ids=["from_fp32", "from_fp16"])
def test_rgb_space_to_depth_fold(dev, shape, offset, src):
    """tlxmi_nchw_to_nhwc_s2d with b = 2 (the RGB stems: one load per pair of horizontal neighbours when the image is 3-channel and
    pair-aligned — 8 bytes of an fp32 image, 4 of an fp16 one — the generic kernel otherwise): channel (ph*2 + pw)*C + c of output
    pixel (h2, w2) = x[n, c, 2*h2+ph, 2*w2+pw]."""
    N, Cc, H, W = shape
    g = torch.Generator().manual_seed(41)
    flat = torch.randn(N * Cc * H * W + 1, generator=g).to(src)
    x = flat[offset: offset + N * Cc * H * W].view(N, Cc, H, W)
    want = x.view(N, Cc, H // 2, 2, W // 2, 2).permute(0, 2, 4, 3, 5, 1).reshape(N, H // 2, W // 2, 4 * Cc)
    xd = flat.to(dev)[offset: offset + N * Cc * H * W].view(N, Cc, H, W)
    got = E.nchw_to_nhwc_s2d(xd, 2, torch.float16)
    torch.cuda.synchronize()
    assert got.shape[:3] == (N, H // 2, W // 2) and got.shape[3] >= 4 * Cc
    torch.testing.assert_close(got[..., :4 * Cc].float().cpu(), want.half().float(), atol=0, rtol=0)
    assert (got[..., 4 * Cc:] == 0).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
@pytest.mark.parametrize("cfg", [(2, 64, 14, 14, 3, 2, 1), (1, 128, 9, 15, 3, 1, 1), (2, 256, 8, 8, 2, 2, 0), (1, 32, 7, 7, 3, 2, 1)],
                         ids=lambda c: "x".join(map(str, c)))
def test_avgpool2d_counts_the_zero_padding(dev, dtype, cfg):
    """nn.AvgPool2d of resnest.py:212-218, 250-256, 271-286 = F.avg_pool2d (padding included in the divisor)."""
    N, Cc, H, W, k, s, p = cfg
    rng = np.random.default_rng(47)
    x = rnd(rng, (N, Cc, H, W))
    if dtype == torch.float16:
        x = q16(x)
    want = torch.nn.functional.avg_pool2d(x, k, s, p)
    got = E.avgpool2d(nchw_to_engine(x, dtype, dev), k, s, p)
    torch.cuda.synchronize()
    torch.testing.assert_close(engine_to_nchw(got), want, **tol(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
@pytest.mark.parametrize("cfg", [(2, 64, 2, 1, 10, 12), (1, 128, 2, 4, 7, 7), (3, 32, 1, 1, 5, 9), (1, 64, 4, 2, 6, 6), (2, 48, 1, 3, 4, 4)],
                         ids=lambda c: "x".join(map(str, c)))
def test_split_attention_and_radix_gap(dev, dtype, cfg):
    """SplatConv.forward resnest.py:149-165 after conv1, with rSoftmax :65-81 restated in torch (the oracle's _rs_splat body)."""
    N, Cc, radix, card, H, W = cfg
    rng = np.random.default_rng(53)
    x = rnd(rng, (N, radix * Cc, H, W))
    logit = rnd(rng, (N, radix * Cc), 2.0)
    if dtype == torch.float16:
        x, logit = q16(x), q16(logit)
    # gap
    splits = torch.chunk(x, radix, dim=1)
    want_gap = sum(splits[1:], splits[0]).mean(dim=(2, 3))
    # attention
    if radix > 1:
        a = logit.reshape(N, card, radix, Cc // card).transpose(1, 2)
        a = torch.softmax(a, dim=1).reshape(N, radix * Cc, 1, 1)
        want = sum(t * s_ for t, s_ in zip(torch.chunk(a, radix, dim=1), splits))
    else:
        want = x * torch.sigmoid(logit).reshape(N, Cc, 1, 1)
    xe = nchw_to_engine(x, dtype, dev)
    got_gap = E.radix_gap(xe, radix)
    got = E.split_attention(xe, logit.to(dtype).to(dev), radix, card)
    torch.cuda.synchronize()
    torch.testing.assert_close(got_gap.float().cpu(), want_gap, **tol(dtype))
    torch.testing.assert_close(engine_to_nchw(got), want, **tol(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
@pytest.mark.parametrize("shape,b", [((2, 3, 224, 224), 16), ((1, 3, 64, 96), 4), ((3, 3, 32, 48), 16), ((1, 5, 24, 24), 4),
                                     ((1, 3, 384, 384), 16)], ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else str(v))
def test_patch_space_to_depth_fold(dev, dtype, shape, b):
    """tlxmi_nchw_to_nhwc_s2d for the patch-embedding folds (vision_transformer.py:197-220 b = 16, swin_transformer.py:490 b = 4):
    channel (ph*b + pw)*C + c of output pixel (h2, w2) = x[n, c, b*h2 + ph, b*w2 + pw]; padding channels are zero."""
    N, Cc, H, W = shape
    g = torch.Generator().manual_seed(61)
    x = torch.randn(shape, generator=g)
    want = x.view(N, Cc, H // b, b, W // b, b).permute(0, 2, 4, 3, 5, 1).reshape(N, H // b, W // b, b * b * Cc)
    got = E.nchw_to_nhwc_s2d(x.to(dev), b, dtype)
    torch.cuda.synchronize()
    assert got.shape[:3] == (N, H // b, W // b) and got.shape[3] >= b * b * Cc
    torch.testing.assert_close(got[..., :b * b * Cc].float().cpu(), want.to(dtype).float(), atol=0, rtol=0)
    assert (got[..., b * b * Cc:] == 0).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
@pytest.mark.parametrize("shape", [(3, 40, 7, 9), (2, 96, 28, 28), (1, 8, 1, 1)], ids=lambda s: "x".join(map(str, s)))
def test_scale_channels(dev, dtype, shape):
    """Squeeze-Excitation gate (mobilenetv3.py:47-56, efficientnet.py:176-178): y[n, c, h, w] = x[n, c, h, w] * s[n, c]."""
    N, Cc, H, W = shape
    rng = np.random.default_rng(67)
    x, sc = rnd(rng, (N, Cc, H, W)), rnd(rng, (N, Cc))
    if dtype == torch.float16:
        x, sc = q16(x), q16(sc)
    got = E.scale_channels(nchw_to_engine(x, dtype, dev), sc.to(dtype).to(dev))
    torch.cuda.synchronize()
    torch.testing.assert_close(engine_to_nchw(got), x * sc[:, :, None, None], **tol(dtype))


def test_swin_module_level_window_helpers_match_the_reference_definitions(dev):
    """swin_transformer.py:85-116: window_partition / window_reverse as module-level functions with the reference's signatures
    (a drop-in import can name them), against the reshape / transpose definitions, both precisions; drop_path is the identity
    in eval mode (:35-47)."""
    from tlxcv_amd.models.classification import swin_transformer as S
    rng = np.random.default_rng(5)
    for dtype in (torch.float16, torch.float32):
        B, H, W, Cc, ws = 3, 14, 21, 32, 7
        x = torch.from_numpy(rng.standard_normal((B, H, W, Cc)).astype(np.float32)).to(dtype)
        ref = x.reshape(B, H // ws, ws, W // ws, ws, Cc).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, Cc)
        win = S.window_partition(x.to(dev), ws)
        assert win.shape == ref.shape and torch.equal(win.cpu(), ref)
        back = S.window_reverse(win, ws, H, W, Cc)
        assert back.shape == x.shape and torch.equal(back.cpu(), x)
    assert S.drop_path(x, 0.3, False) is x and S.DropPath(0.2).set_eval()(x) is x


@pytest.mark.parametrize("shift", [0, 3])
@pytest.mark.parametrize("geom", [(3, 14, 14, 16, 32, 7), (2, 28, 28, 8, 32, 7), (5, 7, 7, 4, 64, 7), (2, 8, 16, 3, 96, 4), (1, 56, 56, 4, 32, 7)],
                         ids=lambda g: "x".join(map(str, g)))
def test_window_attention_on_image_order_rows(dev, geom, shift):
    """tlxmi_attention_windows (round 5): the attention of swin_transformer.py:192-229 with the roll, window_partition, window_reverse
    and roll back of :316-333 as the kernel's row arithmetic — qkv and the result stay in IMAGE order.  Against the oracle's
    roll -> swin_window_partition -> attention -> swin_window_reverse -> roll on the fp16-rounded qkv, with the relative-position
    bias and (shifted blocks) the -100 mask of :288-305; windows of 7 x 7 and 4 x 4, head dims 32 / 64 / 96, one window per image
    (stage 4: no shift there), odd image counts."""
    B, H, W, heads, hd, ws = geom
    if ws >= min(H, W):
        shift = 0
    shift = min(shift, ws - 1)
    rng = np.random.default_rng(B * 131 + H)
    C = heads * hd
    qkv = q16(rnd(rng, (B, H * W, 3 * C)))
    bias = rnd(rng, (heads, ws * ws, ws * ws), 0.5)
    mask = OF.swin_attn_mask(H, W, ws, shift) if shift > 0 else None
    # oracle: image order -> shifted windows -> attention -> back
    x = qkv.reshape(B, H, W, 3 * C)
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = OF.swin_window_partition(x, ws).reshape(-1, ws * ws, 3 * C)
    aw = _ref_attention(xw, heads, hd ** -0.5, bias, mask).reshape(-1, ws, ws, C)
    y = OF.swin_window_reverse(aw, ws, H, W, C)
    if shift > 0:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
    want = y.reshape(B, H * W, C)
    tab = E.attention_table(bias.to(dev), mask.to(dev) if mask is not None else None, ws * ws)
    got = E.attention_windows(qkv.half().to(dev), heads, hd ** -0.5, tab, 0 if mask is None else mask.shape[0], H, W, ws, shift)
    torch.testing.assert_close(got.float().cpu(), want, atol=4e-3, rtol=4e-3)
    # and equal, bit for bit, to the window-order kernel fed by the partition pass (same arithmetic, other addresses)
    win = E.window_partition(qkv.half().to(dev).view(B, H, W, 3 * C), ws, shift)
    old = E.attention_comb(win, heads, hd ** -0.5, tab, 0 if mask is None else mask.shape[0])
    back = E.window_reverse(old, B, H, W, ws, shift)
    assert torch.equal(back.view(B, H * W, C), got)
