"""The fused bottleneck seam (tlxmi_bottleneck_seam, block_seam.hip): conv3 + bn3 + skip + relu of block b and conv1 + bn1 +
relu of block b + 1 (resnet.py:142-156) as one launch — against the oracle and against the two-launch path, for every
compiled channel triple, with row counts that leave partial waves and partial workgroups."""
import numpy as np
import pytest
import torch

from oracle import functional as OF
from tlxcv_amd import engine as E
from util import rnd, q16

pytestmark = pytest.mark.gpu

# (K1, N1, N2, N, H, W): ResNet-50 layer1 / seam into layer2 / layer2 / seam into layer3 / layer3, small extents
# ... / layer3 (256 -> 1024 -> 256 at 14 x 14: engine option "seam256", on by default) incl. a ragged row count (3 x 5 x 7 = 105
# pixels: partial 16-pixel blocks, partial waves, a partial workgroup)
# ... / the triples ResNeXt-50 32x4d and ResNeSt-50 hand to the same kernels (resnext.py:83-117: 128 -> 256 -> 128 inside stage 1,
# 128 -> 256 -> 256 into stage 2, 256 -> 512 -> 256 inside stage 2; resnest.py's bottleneck: 64 / 128 / 256 -> 4x -> next width):
# N1 values the ResNet cases do not reach, each with a ragged row count too
CASES = [(64, 256, 64, 1, 9, 9), (64, 256, 64, 3, 14, 14), (64, 256, 128, 2, 7, 5), (128, 512, 128, 1, 11, 13),
         (128, 512, 256, 2, 6, 6), (256, 1024, 256, 2, 14, 14), (256, 1024, 256, 3, 5, 7),
         (128, 256, 128, 2, 14, 14), (128, 256, 128, 1, 9, 7), (128, 256, 256, 2, 7, 9), (256, 512, 256, 2, 14, 14),
         (256, 512, 256, 1, 5, 11), (256, 1792, 256, 1, 7, 7), (64, 64, 64, 2, 9, 5), (128, 2048, 256, 1, 6, 5)]


def _make(K1, N1, N2, N, H, W, seed):
    rng = np.random.default_rng(seed)
    t2 = q16(torch.relu(rnd(rng, (N, K1, H, W))))
    skip = q16(torch.relu(rnd(rng, (N, N1, H, W))))
    w3 = q16(rnd(rng, (N1, K1, 1, 1), (2.0 / K1) ** 0.5))
    w1 = q16(rnd(rng, (N2, N1, 1, 1), (2.0 / N1) ** 0.5))
    s3 = torch.from_numpy(rng.uniform(0.2, 0.6, N1).astype(np.float32))
    h3 = rnd(rng, (N1,), 0.2)
    s1 = torch.from_numpy(rng.uniform(0.5, 1.5, N2).astype(np.float32))
    h1 = rnd(rng, (N2,), 0.2)
    return t2, skip, w3, w1, s3, h3, s1, h1


@pytest.mark.parametrize("cfg", CASES, ids=lambda c: "x".join(map(str, c)))
def test_seam_matches_the_oracle_and_the_two_launch_path(dev, fp16_mode, cfg):
    K1, N1, N2, N, H, W = cfg
    assert E.bottleneck_seam_supported(K1, N1, N2, torch.float16)
    t2, skip, w3, w1, s3, h3, s1, h1 = _make(*cfg, seed=K1 + N2 + H)
    y_ref = OF.conv_bn_act(t2, w3, s3, h3, skip, E.ACT_RELU)                       # fp32 oracle on the fp16-rounded inputs
    t1_ref = OF.conv_bn_act(q16(y_ref), w1, s1, h1, None, E.ACT_RELU)              # the reduce conv sees y as stored (fp16)
    nh = lambda a: a.permute(0, 2, 3, 1).contiguous().half().to(dev)               # noqa: E731
    pk3, pk1 = E.PackedFilter(w3.to(dev), torch.float16), E.PackedFilter(w1.to(dev), torch.float16)
    s3d, h3d, s1d, h1d = (v.to(dev) for v in (s3, h3, s1, h1))
    y, t1 = E.bottleneck_seam(nh(t2), pk3, s3d, h3d, nh(skip), pk1, s1d, h1d)
    y2 = E.conv2d(nh(t2), pk3, 1, 0, 1, s3d, h3d, nh(skip), E.ACT_RELU)
    t12 = E.conv2d(y2, pk1, 1, 0, 1, s1d, h1d, None, E.ACT_RELU)
    torch.cuda.synchronize()
    assert y.shape == (N, H, W, N1) and t1.shape == (N, H, W, N2)
    torch.testing.assert_close(y.float().cpu().permute(0, 3, 1, 2), y_ref, atol=2e-3, rtol=2e-3)
    torch.testing.assert_close(t1.float().cpu().permute(0, 3, 1, 2), t1_ref, atol=4e-3, rtol=4e-3)
    # against the same engine run as two launches: same fp32 accumulation of fp16 products, possibly another order
    torch.testing.assert_close(y.float(), y2.float(), atol=2e-3, rtol=2e-3)
    torch.testing.assert_close(t1.float(), t12.float(), atol=4e-3, rtol=4e-3)


def test_supported_never_promises_a_launch_that_is_refused(dev, fp16_mode):
    """Every (K1, N1, N2) `tlxmi_bottleneck_seam_supported` answers yes to must launch (256 -> 2048 -> 256 used to be promised
    and then refused for 165888 bytes of LDS); the widest N1 per kernel is launched on a handful of pixels."""
    for K1, N2 in ((64, 64), (64, 128), (128, 128), (128, 256), (256, 256)):
        widest = max(n1 for n1 in range(64, 4097, 64) if E.bottleneck_seam_supported(K1, n1, N2, torch.float16))
        assert not E.bottleneck_seam_supported(K1, widest + 64, N2, torch.float16)
        t2, skip, w3, w1, s3, h3, s1, h1 = _make(K1, widest, N2, 1, 3, 5, seed=widest)
        nh = lambda a: a.permute(0, 2, 3, 1).contiguous().half().to(dev)               # noqa: E731
        pk3, pk1 = E.PackedFilter(w3.to(dev), torch.float16), E.PackedFilter(w1.to(dev), torch.float16)
        y, t1 = E.bottleneck_seam(nh(t2), pk3, s3.to(dev), h3.to(dev), nh(skip), pk1, s1.to(dev), h1.to(dev))
        y_ref = OF.conv_bn_act(t2, w3, s3, h3, skip, E.ACT_RELU)
        torch.testing.assert_close(y.float().cpu().permute(0, 3, 1, 2), y_ref, atol=2e-3, rtol=2e-3)
    assert not E.bottleneck_seam_supported(256, 2048, 256, torch.float16)


def test_seam_at_full_size_is_consistent_with_two_launches(dev, fp16_mode):
    """BASELINE-sized: 64 images at 56 x 56 (200704 pixels): the fused launch and the two convolutions agree everywhere."""
    cfg = (64, 256, 64, 64, 56, 56)
    t2, skip, w3, w1, s3, h3, s1, h1 = _make(*cfg, seed=99)
    nh = lambda a: a.permute(0, 2, 3, 1).contiguous().half().to(dev)               # noqa: E731
    pk3, pk1 = E.PackedFilter(w3.to(dev), torch.float16), E.PackedFilter(w1.to(dev), torch.float16)
    s3d, h3d, s1d, h1d = (v.to(dev) for v in (s3, h3, s1, h1))
    a, b = nh(t2), nh(skip)
    y, t1 = E.bottleneck_seam(a, pk3, s3d, h3d, b, pk1, s1d, h1d)
    y2 = E.conv2d(a, pk3, 1, 0, 1, s3d, h3d, b, E.ACT_RELU)
    t12 = E.conv2d(y2, pk1, 1, 0, 1, s1d, h1d, None, E.ACT_RELU)
    torch.cuda.synchronize()
    assert (y.float() - y2.float()).abs().max().item() <= 4e-3 and (t1.float() - t12.float()).abs().max().item() <= 8e-3


@pytest.mark.parametrize("N,H,W", [(1, 9, 9), (2, 14, 14), (16, 56, 56)])
def test_seam_with_the_projection_shortcut_inside(dev, fp16_mode, N, H, W):
    """layer1.0 (resnet.py:246-261): skip = bn_d(conv_d(x)) computed in the seam launch; equal to the three-launch path
    (shortcut conv, expand conv + skip, reduce conv) to fp16 rounding, and to the oracle."""
    K1, N1, N2 = 64, 256, 64
    t2, _, w3, w1, s3, h3, s1, h1 = _make(K1, N1, N2, N, H, W, seed=5 + H)
    rng = np.random.default_rng(77 + H)
    x = q16(torch.relu(rnd(rng, (N, K1, H, W))))
    wd = q16(rnd(rng, (N1, K1, 1, 1), (2.0 / K1) ** 0.5))
    sd = torch.from_numpy(rng.uniform(0.5, 1.5, N1).astype(np.float32))
    hd = rnd(rng, (N1,), 0.2)
    skip_ref = q16(OF.conv_bn_act(x, wd, sd, hd, None, E.ACT_NONE))                  # the shortcut as stored: fp16
    y_ref = OF.conv_bn_act(t2, w3, s3, h3, skip_ref, E.ACT_RELU)
    t1_ref = OF.conv_bn_act(q16(y_ref), w1, s1, h1, None, E.ACT_RELU)
    nh = lambda a: a.permute(0, 2, 3, 1).contiguous().half().to(dev)                 # noqa: E731
    pk3, pk1, pkd = (E.PackedFilter(w.to(dev), torch.float16) for w in (w3, w1, wd))
    dv = lambda *ts: [t.to(dev) for t in ts]                                         # noqa: E731
    s3d, h3d, s1d, h1d, sdd, hdd = dv(s3, h3, s1, h1, sd, hd)
    y, t1 = E.bottleneck_seam(nh(t2), pk3, s3d, h3d, nh(x), pk1, s1d, h1d, proj=(pkd, sdd, hdd))
    skip2 = E.conv2d(nh(x), pkd, 1, 0, 1, sdd, hdd, None, E.ACT_NONE)
    y2 = E.conv2d(nh(t2), pk3, 1, 0, 1, s3d, h3d, skip2, E.ACT_RELU)
    t12 = E.conv2d(y2, pk1, 1, 0, 1, s1d, h1d, None, E.ACT_RELU)
    torch.cuda.synchronize()
    torch.testing.assert_close(y.float().cpu().permute(0, 3, 1, 2), y_ref, atol=3e-3, rtol=3e-3)
    torch.testing.assert_close(t1.float().cpu().permute(0, 3, 1, 2), t1_ref, atol=5e-3, rtol=5e-3)
    torch.testing.assert_close(y.float(), y2.float(), atol=3e-3, rtol=3e-3)
    torch.testing.assert_close(t1.float(), t12.float(), atol=5e-3, rtol=5e-3)


def test_unsupported_triples_are_reported_not_run(dev):
    assert not E.bottleneck_seam_supported(96, 384, 96, torch.float16)
    assert not E.bottleneck_seam_supported(64, 256, 64, torch.float32)


@pytest.mark.parametrize("rows,hidden", [(4096 + 37, 512), (200704, 512), (8192, 256), (5000, 1024)], ids=lambda v: str(v))
def test_mlp_as_one_launch(dev, rows, hidden):
    """tlxmi_mlp_seam (round 5; swin_transformer.py:62-82 Mlp + the residual of :335): out = fc2(gelu(fc1(x) + b1)) + b2 + res with the
    hidden activations kept on the CU (the seam kernel's MLP form, 128 -> hidden -> 128).  Against torch fp32 (exact-erf GELU) on the
    fp16-rounded operands, against the two-launch path (fc1 + GELU, fc2 + residual) on the same inputs, in place (out = res), a ragged
    last workgroup, Swin-B stage 1's row count at half batch 64."""
    import numpy as np
    from tlxcv_amd import engine as E
    from util import rnd, q16
    rng = np.random.default_rng(rows % 1000 + hidden)
    K = N = 128
    x = q16(rnd(rng, (rows, K)))
    res = q16(rnd(rng, (rows, N)))
    w1 = q16(rnd(rng, (hidden, K), (1.0 / K) ** 0.5))
    b1 = rnd(rng, (hidden,), 0.3)
    w2 = q16(rnd(rng, (N, hidden), (1.0 / hidden) ** 0.5))
    b2 = rnd(rng, (N,), 0.2)
    assert E.mlp_seam_supported(rows, K, hidden, N, torch.float16)
    pk1, pk2 = E.PackedFilter(w1.to(dev), torch.float16), E.PackedFilter(w2.to(dev), torch.float16)
    xd, rd = x.half().to(dev), res.half().to(dev)
    got = E.mlp_seam(xd, pk1, b1.to(dev), pk2, b2.to(dev), rd)
    sel = torch.cat([torch.arange(0, 600), torch.arange(rows - 600, rows)])
    h = torch.nn.functional.gelu(x[sel] @ w1.t() + b1)
    want = q16(h) @ w2.t() + b2 + res[sel]              # the hidden activations are rounded to fp16 between the two products, as in two launches
    torch.testing.assert_close(got[sel.to(dev)].float().cpu(), want, atol=6e-3, rtol=6e-3)
    two = E.linear(E.linear(xd, pk1, b1.to(dev), act=E.ACT_GELU), pk2, b2.to(dev), res=rd)
    torch.testing.assert_close(got.float(), two.float(), atol=6e-3, rtol=6e-3)
    inplace = rd.clone()
    out = E.mlp_seam(xd, pk1, b1.to(dev), pk2, b2.to(dev), inplace, out=inplace)
    assert out.data_ptr() == inplace.data_ptr() and torch.equal(out, got)
    assert torch.equal(E.mlp_seam(xd, pk1, b1.to(dev), pk2, b2.to(dev), rd), got)
