"""Reads of LDS bytes a kernel has not written itself.

Every big kernel here stages its operands by LDS-DMA behind hand-counted `s_waitcnt vmcnt(N)` and raw `s_barrier`s.  A wait
that is one count short, or a fragment read one barrier early, returns whatever the LDS held before — and in every
repeated-launch test that is the previous launch's IDENTICAL tile or table, so the result is right by accident ("two forwards
bit-identical" cannot see it; round 3 lost a kernel variant to exactly this).  Method: a probe kernel (tests/probe/poison.hip,
built by `make tune` into tests/probe/libpoison.so — test infrastructure, not product) fills all 160 KiB of every CU's LDS with
a pattern right before EVERY compute launch of a forward on the PRODUCT library; the result must equal the unpoisoned
forward bit for bit, for three patterns: fp16 NaN pairs (a stale operand tile), fp32 +infinity (a stale scale / shift / bias /
mask table entry; survives ReLU), and fp16 1.0 pairs = fp32 0.0078 (a finite value that changes sums without tripping
isfinite).  Run over the three bench configurations at their bench sizes — the shipped dispatch: gemm_stream with and without
a residual, gemm_pp plain / CONV / 128-row tiles, gemm256, gemm_wreg, conv_halo (+ pooled stem), the block seams, window
attention with the resident table, fused attention, LayerNorm — and, op by op, over the layer shapes of those models, so a
failure names the launch."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from conftest import REPO
from tlxcv_amd import seeded

pytestmark = pytest.mark.gpu

PATTERNS = [("fp16 NaN pairs", 0x7FC07FC0), ("fp32 +inf", 0x7F800000), ("fp16 ones", 0x3C003C00)]


@pytest.fixture(scope="module")
def probe():
    path = os.path.join(REPO, "tests", "probe", "libpoison.so")
    assert os.path.exists(path), f"{path} missing: build it with `make -C tlxcv_amd/csrc tune` (__graft_entry__.build() does)"
    lib = C.CDLL(path)
    lib.poison_lds.argtypes = [C.c_uint, C.c_void_p]
    lib.poison_lds.restype = C.c_int
    return lib


class poisoned:
    """with poisoned(probe, pattern): every status-returning libtlxmi call is preceded by the LDS fill on the same stream."""

    def __init__(self, probe, pattern):
        self.probe, self.pattern, self.launches = probe, pattern, 0

    def __enter__(self):
        from tlxcv_amd import _lib
        self.lib, self.orig = _lib, _lib.call

        def call(name, *args):
            rc = self.probe.poison_lds(C.c_uint(self.pattern), C.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, f"poison launch failed ({rc})"
            self.launches += 1
            return self.orig(name, *args)
        _lib.call = call
        return self

    def __exit__(self, *exc):
        self.lib.call = self.orig
        return False


def _same(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and torch.equal(a.view(torch.int16 if a.dtype == torch.float16 else torch.int32),
                                                                     b.view(torch.int16 if b.dtype == torch.float16 else torch.int32))


def _diff_report(y, ref):
    bad = (y != ref) | (torch.isnan(y) != torch.isnan(ref))
    rows = bad.reshape(bad.shape[0], -1).any(1).nonzero().flatten()
    return f"{int(bad.sum())} elements differ (NaN {int(torch.isnan(y).sum())}, inf {int(torch.isinf(y).sum())}); first rows {rows[:8].tolist()}"


MODELS = [("resnet50", 256, 1), ("vit_base_patch16_224", 256, 2), ("swintransformer_base_patch4_window7_224", 128, 3),
          ("ResNeXt", 32, 4), ("vgg16", 32, 5)]


@pytest.mark.parametrize("ctor,batch,seed", MODELS, ids=[m[0] for m in MODELS])
def test_forward_is_bit_identical_with_poisoned_lds_before_every_launch(dev, fp16_mode, probe, ctor, batch, seed):
    from tlxcv_amd import models
    m = getattr(models, ctor)()
    m.load_dict(seeded.fill(seeded.shapes_of(m), seed))
    m = m.to(dev).set_eval()
    x = torch.from_numpy(seeded.image_batch(32, 99)).repeat(batch // 32, 1, 1, 1).to(dev)
    x += 0.01 * torch.arange(batch, device=dev, dtype=x.dtype).view(-1, 1, 1, 1)      # no two images alike
    ref = m(x).clone()
    ref2 = m(x).clone()
    torch.cuda.synchronize()
    assert _same(ref, ref2) and torch.isfinite(ref).all()
    for name, pat in PATTERNS:
        with poisoned(probe, pat) as p:
            y = m(x).clone()
        torch.cuda.synchronize()
        assert p.launches > 10
        assert _same(y, ref), f"{ctor} batch {batch}, LDS poisoned with {name} before each of {p.launches} launches: " + _diff_report(y.float(), ref.float())


def _lin_cases(dev):
    """Linear layers at the bench shapes of ViT-B/16 (batch 256 / the 128-image halves the forward runs) and Swin-B stage 1-4
    (batch 64 halves): (name, thunk)."""
    from tlxcv_amd import engine as E
    g = torch.Generator().manual_seed(7)
    out = []

    def lin(tag, M, K, N, act=E.ACT_NONE, with_res=False, plan=None):
        x = torch.randn((M, K), generator=g).half().to(dev)
        w = (torch.randn((N, K), generator=g) * K ** -0.5).to(dev)
        b = (torch.randn(N, generator=g) * 0.1).to(dev)
        pk = E.PackedFilter(w, torch.float16)
        res = torch.randn((M, N), generator=g).half().to(dev) if with_res else None

        def run():
            with E.shared_plan(plan):
                return E.linear(x, pk, b, act=act, res=res)
        out.append((f"{tag} M={M} K={K} N={N}" + (" +res" if with_res else "") + (f" plan={plan}" if plan else ""), run))
    for M, plan in ((50432, None), (25216, "half")):
        lin("vit qkv", M, 768, 2304, plan=plan)
        lin("vit proj", M, 768, 768, with_res=True, plan=plan)
        lin("vit fc1+gelu", M, 768, 3072, act=E.ACT_GELU, plan=plan)
        lin("vit fc2", M, 3072, 768, with_res=True, plan=plan)
    for C_, L in ((128, 3136), (256, 784), (512, 196), (1024, 49)):
        M = 64 * L
        lin("swin qkv", M, C_, 3 * C_, plan="full")
        lin("swin proj", M, C_, C_, plan="full")
        lin("swin fc1+gelu", M, C_, 4 * C_, act=E.ACT_GELU, plan="full")
        lin("swin fc2", M, 4 * C_, C_, with_res=True, plan="full")
    lin("odd rows / tail tile", 1000, 768, 520)
    lin("head", 256, 2048, 1000)
    return out


def test_linear_layers_op_by_op_with_poisoned_lds(dev, fp16_mode, probe):
    failures = []
    for name, run in _lin_cases(dev):
        ref = run().clone()
        torch.cuda.synchronize()
        for pname, pat in PATTERNS:
            with poisoned(probe, pat):
                y = run().clone()
            torch.cuda.synchronize()
            if not _same(y, ref):
                failures.append(f"{name} [{pname}]: " + _diff_report(y.float(), ref.float()))
    assert not failures, "\n".join(failures)


def test_convolutions_seams_and_attention_op_by_op_with_poisoned_lds(dev, fp16_mode, probe):
    """ResNet-50's conv shapes at batch 128 (the halves of the bench forward), the block seams, grouped conv, attention."""
    from tlxcv_amd import engine as E
    g = torch.Generator().manual_seed(11)
    cases = []

    def conv(tag, N, H, Cin, Cout, k, stride, with_res=False, groups=1):
        x = torch.relu(torch.randn((N, H, H, Cin), generator=g)).half().to(dev)
        w = (torch.randn((Cout, Cin // groups, k, k), generator=g) * (2.0 / (Cin // groups * k * k)) ** 0.5).to(dev)
        s = (torch.rand(Cout, generator=g) + 0.5).to(dev)
        h = (torch.randn(Cout, generator=g) * 0.1).to(dev)
        Ho = (H + 2 * (k // 2) - k) // stride + 1
        res = torch.randn((N, Ho, Ho, Cout), generator=g).half().to(dev) if with_res else None
        if groups == 1:
            pk = E.PackedFilter(w, torch.float16)
            run = lambda: E.conv2d(x, pk, stride, k // 2, 1, s, h, res, E.ACT_RELU)      # noqa: E731
        else:
            pk = E.PackedGroupFilter(w, groups, torch.float16)
            run = lambda: E.group_conv2d(x, pk, stride, k // 2, 1, s, h, res, E.ACT_RELU)      # noqa: E731

        def planned():
            with E.shared_plan("half"):
                return run()
        cases.append((f"{tag} {Cin}->{Cout} k{k} s{stride} @{H} x{N}" + (" +res" if with_res else ""), planned))
    B = 128
    conv("layer1 3x3", B, 56, 64, 64, 3, 1)
    conv("layer1 expand", B, 56, 64, 256, 1, 1, with_res=True)
    conv("layer1 reduce", B, 56, 256, 64, 1, 1)
    conv("layer2 3x3 s2", B, 56, 128, 128, 3, 2)
    conv("layer2 3x3", B, 28, 128, 128, 3, 1)
    conv("layer2 shortcut", B, 56, 256, 512, 1, 2)
    conv("layer3 3x3", B, 14, 256, 256, 3, 1)
    conv("layer3 3x3 s2", B, 28, 256, 256, 3, 2)
    conv("layer3 expand", B, 14, 256, 1024, 1, 1, with_res=True)
    conv("layer4 3x3", B, 7, 512, 512, 3, 1)
    conv("layer4 expand", B, 7, 512, 2048, 1, 1, with_res=True)
    conv("layer4 reduce", B, 7, 2048, 512, 1, 1)
    conv("resnext g32", 32, 56, 128, 128, 3, 1, groups=32)
    conv("resnext g32 s2", 32, 56, 256, 256, 3, 2, groups=32)

    def seam(K1, N1, N2, N, H):
        t2 = torch.relu(torch.randn((N, H, H, K1), generator=g)).half().to(dev)
        skip = torch.relu(torch.randn((N, H, H, N1), generator=g)).half().to(dev)
        pk3 = E.PackedFilter((torch.randn((N1, K1, 1, 1), generator=g) * (2.0 / K1) ** 0.5).to(dev), torch.float16)
        pk1 = E.PackedFilter((torch.randn((N2, N1, 1, 1), generator=g) * (2.0 / N1) ** 0.5).to(dev), torch.float16)
        s3, h3 = (torch.rand(N1, generator=g) * 0.4 + 0.2).to(dev), (torch.randn(N1, generator=g) * 0.2).to(dev)
        s1, h1 = (torch.rand(N2, generator=g) + 0.5).to(dev), (torch.randn(N2, generator=g) * 0.2).to(dev)
        cases.append((f"seam {K1}->{N1}->{N2} @{H} x{N}", lambda: torch.cat([o.reshape(-1) for o in E.bottleneck_seam(t2, pk3, s3, h3, skip, pk1, s1, h1)])))
    seam(64, 256, 64, 64, 56)
    seam(64, 256, 128, 64, 56)
    seam(128, 512, 128, 64, 28)
    seam(128, 512, 256, 64, 28)
    seam(256, 1024, 256, 128, 14)
    seam(128, 256, 128, 32, 56)
    seam(256, 512, 256, 32, 28)

    qkv = torch.randn((128, 197, 3 * 768), generator=g).half().to(dev)
    cases.append(("vit attention 197 tokens x 12 heads", lambda: E.attention(qkv, 12, 0.125)))

    failures = []
    for name, run in cases:
        ref = run().clone()
        torch.cuda.synchronize()
        for pname, pat in PATTERNS:
            with poisoned(probe, pat):
                y = run().clone()
            torch.cuda.synchronize()
            if not _same(y, ref):
                failures.append(f"{name} [{pname}]: " + _diff_report(y.float(), ref.float()))
    assert not failures, "\n".join(failures)


def test_folded_layernorm_and_window_attention_op_by_op_with_poisoned_lds(dev, fp16_mode, probe):
    """Round 5's kernels one by one at the shapes the ViT-B/16 / Swin-B forwards run them (half batches, both CU plans): tlxmi_linear_stats
    (gemm_stream STATS with and without a residual: its statistics cross the waves through an LDS scratch; gemm_pp LNF for the 8-K-tile
    residual producer of Swin-B stage 3 and the launches of few tiles), tlxmi_linear_ln (gemm_stream ROWAFF, plain and GELU: the
    statistics planes, their conversion to (a, b) and the prefetch all live in LDS), tlxmi_attention_windows.
    Every LDS byte holds the pattern before every launch; outputs and statistics must not move by a bit."""
    from tlxcv_amd import engine as E
    g = torch.Generator().manual_seed(11)
    cases = []

    def fold(tag, M, K, D, N2, act, plan):
        x = torch.randn((M, K), generator=g).half().to(dev)
        w = (torch.randn((D, K), generator=g) * K ** -0.5).to(dev)
        b = (torch.randn(D, generator=g) * 0.1).to(dev)
        r = torch.randn((M, D), generator=g).half().to(dev)
        pk = E.PackedFilter(w, torch.float16)
        prep = E.LinearLN((torch.randn((N2, D), generator=g) * D ** -0.5).to(dev), (torch.randn(N2, generator=g) * 0.1).to(dev),
                          (torch.rand(D, generator=g) + 0.5).to(dev), (torch.randn(D, generator=g) * 0.2).to(dev), torch.float16)

        def producer(res):
            with E.shared_plan(plan):
                y, part = E.linear_stats(x, pk, b, res=res)
            return torch.cat([y.float().reshape(-1), part.reshape(-1)])
        with E.shared_plan(plan):
            y0, part0 = E.linear_stats(x, pk, b, res=r)
        cases.append((f"{tag} producer + res M={M} K={K} N={D} plan={plan}", lambda: producer(r)))
        cases.append((f"{tag} producer M={M} K={K} N={D} plan={plan}", lambda: producer(None)))

        def consumer():
            with E.shared_plan(plan):
                return E.linear_ln(y0, prep, part0, 1e-5, act)
        cases.append((f"{tag} consumer M={M} K={D} N={N2} act={act} plan={plan}", consumer))
    fold("vit proj -> qkv", 25216, 768, 768, 2304, E.ACT_NONE, "full")
    fold("vit fc2 -> fc1", 25216, 3072, 768, 3072, E.ACT_GELU, "half")
    fold("swin-3 proj -> fc1 (8 K tiles: gemm_pp LNF)", 12544, 512, 512, 2048, E.ACT_GELU, "full")
    fold("swin-3 fc2 -> qkv", 12544, 2048, 512, 1536, E.ACT_NONE, "full")
    fold("swin-4 fc2 -> qkv", 3136, 4096, 1024, 3072, E.ACT_NONE, "full")

    for (B, H, W, heads, shift) in ((64, 14, 14, 16, 3), (64, 28, 28, 8, 0), (64, 7, 7, 32, 0)):
        C = heads * 32
        qkv = torch.randn((B, H * W, 3 * C), generator=g).half().to(dev)
        bias = (torch.randn((heads, 49, 49), generator=g) * 0.5).to(dev)
        from oracle import functional as OF
        mask = OF.swin_attn_mask(H, W, 7, shift).to(dev) if shift else None
        tab = E.attention_table(bias, mask, 49)
        cases.append((f"window attention on image rows {H}x{W} heads {heads} shift {shift}",
                      lambda qkv=qkv, tab=tab, mask=mask, H=H, W=W, heads=heads, shift=shift: E.attention_windows(qkv, heads, 32 ** -0.5, tab, 0 if mask is None else mask.shape[0], H, W, 7, shift)))
    failures = []
    for name, run in cases:
        ref = run().clone()
        torch.cuda.synchronize()
        for pname, pat in PATTERNS:
            with poisoned(probe, pat):
                y = run().clone()
            torch.cuda.synchronize()
            if not _same(y, ref):
                failures.append(f"{name} [{pname}]: " + _diff_report(y.float(), ref.float()))
    assert not failures, "\n".join(failures)
