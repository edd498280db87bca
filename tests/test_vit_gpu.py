"""ViT parity on the MI355X against golden logits produced by the reference's own
vision_transformer.py (oracle/gen_golden.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from util import check_fp16_logits
from tlxcv_amd import seeded

pytestmark = pytest.mark.gpu


def build(arch, seed, dev):
    from tlxcv_amd import models
    m = getattr(models, arch)()
    m.load_dict(seeded.fill(seeded.shapes_of(m), seed))
    return m.to(dev).set_eval()


@pytest.mark.parametrize("fname", ["vit_b16_b2.npz", "vit_small_b1.npz", "vit_b16_384_b1.npz"])
def test_fp32_matches_golden_1e4_and_argmax_exact(dev, fp32_mode, fname):
    g = np.load(os.path.join(GOLDEN, fname))
    m = build(str(g["arch"]), int(g["weight_seed"]), dev)
    hw = int(g["hw"]) if "hw" in g.files else 224
    x = torch.from_numpy(seeded.image_batch(int(g["batch"]), int(g["input_seed"]), hw=hw)).to(dev)
    y = m(x)
    err = np.abs(y.cpu().numpy() - g["logits"]).max()
    assert err <= 1e-4, err
    from tlxcv_amd.tasks import ImageClassification
    assert (ImageClassification(m).predict(x).cpu().numpy() == g["argmax"]).all()


def test_fp16_tracks_golden(dev, fp16_mode):
    g = np.load(os.path.join(GOLDEN, "vit_b16_b2.npz"))
    m = build("vit_base_patch16_224", int(g["weight_seed"]), dev)
    x = torch.from_numpy(seeded.image_batch(2, int(g["input_seed"]))).to(dev)
    y = m(x).float().cpu().numpy()
    ref = g["logits"]
    check_fp16_logits(y, ref, g["argmax"], "vit_b16_b2")


def test_fp16_384_tracks_golden(dev, fp16_mode):
    """vit_base_patch16_384 (vision_transformer.py:358-, 577 tokens) in the throughput mode against the reference-file fixture."""
    g = np.load(os.path.join(GOLDEN, "vit_b16_384_b1.npz"))
    m = build("vit_base_patch16_384", int(g["weight_seed"]), dev)
    x = torch.from_numpy(seeded.image_batch(1, int(g["input_seed"]), hw=384)).to(dev)
    y = m(x).float().cpu().numpy()
    check_fp16_logits(y, g["logits"], g["argmax"], "vit_b16_384_b1")


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_patch_embedding_as_one_linear_equals_the_conv_path(dev, prec):
    """The two arms of forward_features' patch embedding — tlxmi_patchify + ONE Linear over all token rows with the per-row
    residual [cls + pos[0] - bias | pos[1:]] (default), and the space-to-depth conv writing rows 1.. of each image + the cls row —
    against the golden logits and against each other."""
    import tlxcv_amd
    from tlxcv_amd import engine as E
    tlxcv_amd.set_precision(prec)
    try:
        g = np.load(os.path.join(GOLDEN, "vit_b16_b2.npz"))
        m = build("vit_base_patch16_224", int(g["weight_seed"]), dev)
        x = torch.from_numpy(seeded.image_batch(2, int(g["input_seed"]))).to(dev)
        ys = {}
        for arm in (True, False):
            E.set_option("patch_linear", arm)
            ys[arm] = m(x).float().cpu().numpy()
        tol = 1e-4 if prec == "fp32" else 3e-2
        assert np.abs(ys[True] - ys[False]).max() <= tol
        assert np.abs(ys[True] - g["logits"]).max() <= tol and np.abs(ys[False] - g["logits"]).max() <= tol
        assert (ys[True].argmax(-1) == g["argmax"]).all() and (ys[False].argmax(-1) == g["argmax"]).all()
    finally:
        E.set_option("patch_linear", True)
        tlxcv_amd.set_precision("fp16")


def test_wrong_image_size_asserts(dev, fp16_mode):
    m = build("vit_base_patch16_224", 2, dev)
    with pytest.raises(AssertionError, match="doesn't match"):      # vision_transformer.py:217-219
        m(torch.zeros(1, 3, 192, 192, device=dev))


def test_vit_384_runs_the_long_sequence_attention(dev, fp32_mode):
    """vit_base_patch16_384 (vision_transformer.py:209, 577 tokens) against the oracle restatement, fp32."""
    from oracle import functional as OF
    from tlxcv_amd import models
    m = models.vit_base_patch16_384()
    params = seeded.fill(seeded.shapes_of(m), 3)
    m.load_dict(params)
    m = m.to(dev).set_eval()
    x = torch.from_numpy(seeded.image_batch(1, 5, hw=384))
    with torch.no_grad():
        ref = OF.vit({k: torch.from_numpy(v) for k, v in params.items()}, x, "vit_base_patch16_384").numpy()
    y = m(x.to(dev)).cpu().numpy()
    assert np.abs(y - ref).max() <= 1e-4
    assert (y.argmax(1) == ref.argmax(1)).all()


@pytest.mark.parametrize("batch", [16, 64])
def test_layernorm_folded_into_the_gemms_tracks_golden(dev, fp16_mode, batch):
    """Round 5: norm1 / norm2 of every block (vision_transformer.py:172-175) without a LayerNorm launch — statistics out of the
    proj / fc2 / patch-embedding epilogues, the normalisation in the qkv / fc1 epilogues (engine option "lnfold").  The golden
    images of vit_b16_b2 planted in a filler batch big enough for the folded path (>= 2048 token rows; 64: the two-stream forward):
    both arms against the reference-file logits, and against each other."""
    from tlxcv_amd import engine as E
    g = np.load(os.path.join(GOLDEN, "vit_b16_b2.npz"))
    m = build("vit_base_patch16_224", int(g["weight_seed"]), dev)
    gold = seeded.image_batch(2, int(g["input_seed"]))
    x = seeded.image_batch(batch, 123)
    rows = [1, batch - 1]
    x[rows] = gold
    x = torch.from_numpy(x).to(dev)
    ys, launches = {}, {}
    keep = E.option_value("lnfold_min_rows_one_stream")
    E.set_option("lnfold_min_rows_one_stream", 0)      # (the product folds a one-stream forward from 12 k rows only: here the path itself is under test)
    try:
        for arm in (True, False):
            E.set_option("lnfold", arm)
            n0 = _count_ln(m)
            ys[arm] = m(x).float().cpu().numpy()
            launches[arm] = _count_ln(m) - n0
    finally:
        E.set_option("lnfold", True)
        E.set_option("lnfold_min_rows_one_stream", keep)
    assert launches[True] == 0 and launches[False] >= 24 * (2 if batch >= 64 else 1)      # the folded arm ran no LayerNorm module
    for arm in (True, False):
        check_fp16_logits(ys[arm][rows], g["logits"], g["argmax"], "vit_b16_b2")
    span = float(g["logits"].max() - g["logits"].min())
    assert np.abs(ys[True] - ys[False]).max() <= 0.006 * span


_LN_CALLS = [0]


def _count_ln(m):
    """Calls of nn.LayerNorm.forward so far (the folded path never calls the module)."""
    from tlxcv_amd.tlx import nn
    if not getattr(nn.LayerNorm, "_counted", False):
        fwd = nn.LayerNorm.forward

        def counted(self, x):
            _LN_CALLS[0] += 1
            return fwd(self, x)
        nn.LayerNorm.forward = counted
        nn.LayerNorm._counted = True
    return _LN_CALLS[0]
