"""The oracle against the committed golden fixtures (CPU).  The fixtures were produced by running
the reference's own resnet.py on the oracle's tensorlayerx stand-in (oracle/gen_golden.py); here the
restatement alone must reproduce them from the recorded seeds on whatever torch build is present."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import functional as OF
from tlxcv_amd import seeded


def _params(model_ctor, seed):
    m = model_ctor()
    return {k: torch.from_numpy(v) for k, v in seeded.fill(seeded.shapes_of(m), seed).items()}


@pytest.mark.parametrize("fname,depth", [("resnet18_b2.npz", 18), ("resnet50_b4.npz", 50)])
def test_resnet_restatement_reproduces_golden(fname, depth):
    from tlxcv_amd import models
    g = np.load(os.path.join(GOLDEN, fname))
    p = _params(getattr(models, f"resnet{depth}"), int(g["weight_seed"]))
    x = torch.from_numpy(seeded.image_batch(int(g["batch"]), int(g["input_seed"])))
    with torch.no_grad():
        y = OF.resnet(p, x, depth)
    assert np.abs(y.numpy() - g["logits"]).max() <= 1e-4
    assert (y.argmax(-1).numpy() == g["argmax"]).all()
    assert str(g["pinned_by"]) == "reference-file-on-tlx_cpu"


def test_fold_bn_equals_batch_norm():
    rng = np.random.default_rng(0)
    C = 16
    g, b, m = (torch.from_numpy(rng.standard_normal(C).astype(np.float32)) for _ in range(3))
    v = torch.from_numpy(rng.uniform(0.5, 2, C).astype(np.float32))
    x = torch.from_numpy(rng.standard_normal((2, C, 5, 5)).astype(np.float32))
    s, sh = OF.fold_bn(g, b, m, v, 1e-5)
    ref = torch.nn.functional.batch_norm(x, m, v, g, b, False, 0.0, 1e-5)
    assert (x * s.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) - ref).abs().max() < 1e-5


@pytest.mark.parametrize("fname", ["vit_b16_b2.npz", "vit_small_b1.npz"])
def test_vit_restatement_reproduces_golden(fname):
    from tlxcv_amd import models
    g = np.load(os.path.join(GOLDEN, fname))
    arch = str(g["arch"])
    p = _params(getattr(models, arch), int(g["weight_seed"]))
    x = torch.from_numpy(seeded.image_batch(int(g["batch"]), int(g["input_seed"])))
    with torch.no_grad():
        y = OF.vit(p, x, arch)
    assert np.abs(y.numpy() - g["logits"]).max() <= 1e-4
    assert (y.argmax(-1).numpy() == g["argmax"]).all()
