"""End-to-end ResNet parity on the MI355X against the committed golden logits (which come from the
reference's own resnet.py, see oracle/gen_golden.py) and against the live oracle restatement."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from util import check_fp16_logits
from oracle import functional as OF
from tlxcv_amd import seeded

pytestmark = pytest.mark.gpu


def build(name, seed, dev, **kw):
    from tlxcv_amd import models
    m = getattr(models, name)(**kw)
    params = seeded.fill(seeded.shapes_of(m), seed)
    m.load_dict(params)
    return m.to(dev).set_eval(), {k: torch.from_numpy(v) for k, v in params.items()}


@pytest.mark.parametrize("fname,arch", [("resnet50_b4.npz", "resnet50"), ("resnet18_b2.npz", "resnet18")])
def test_fp32_matches_golden_1e4_and_argmax_exact(dev, fp32_mode, fname, arch):
    g = np.load(os.path.join(GOLDEN, fname))
    m, _ = build(arch, int(g["weight_seed"]), dev)
    x = torch.from_numpy(seeded.image_batch(int(g["batch"]), int(g["input_seed"]))).to(dev)
    y = m(x)
    assert y.dtype == torch.float32 and y.shape == g["logits"].shape
    err = np.abs(y.cpu().numpy() - g["logits"]).max()
    assert err <= 1e-4, err                                   # north_star: 1e-4 fp32
    from tlxcv_amd.tasks import ImageClassification
    pred = ImageClassification(m).predict(x)
    assert pred.dtype == torch.int64
    assert (pred.cpu().numpy() == g["argmax"]).all()          # bit-exact class indices


def test_fp16_tracks_golden(dev, fp16_mode):
    g = np.load(os.path.join(GOLDEN, "resnet50_b4.npz"))
    m, _ = build("resnet50", int(g["weight_seed"]), dev)
    x = torch.from_numpy(seeded.image_batch(4, int(g["input_seed"]))).to(dev)
    y = m(x).float().cpu().numpy()
    ref = g["logits"]
    # fp16 storage between 53 fused layers (bound and argmax rule: tests/util.py); top-5 sets overlap
    check_fp16_logits(y, ref, g["argmax"], "resnet50_b4")
    for i in range(4):
        assert len(set(np.argsort(-y[i])[:5]) & set(np.argsort(-ref[i])[:5])) >= 3


def test_other_input_sizes_and_feature_mode(dev, fp32_mode):
    """with_pool / num_classes variants (resnet.py:295-299) on a non-square, non-224 input."""
    m, p = build("resnet50", 21, dev, num_classes=0, with_pool=False)
    x = torch.from_numpy(seeded.image_batch(1, 5, hw=96))[:, :, :64, :].contiguous()
    y = m(x.to(dev))
    with torch.no_grad():
        ref = OF.resnet(p, x, 50, num_classes=0, with_pool=False)
    assert tuple(y.shape) == tuple(ref.shape) == (1, 2048, 2, 3)
    torch.testing.assert_close(y.cpu(), ref, atol=1e-4, rtol=1e-4)


def test_layerwise_api_equals_fused_graph(dev, fp32_mode):
    """The TensorLayerX-style one-layer-at-a-time calls (conv, bn, relu as separate kernels) and the
    fused graph must agree: this is the drop-in surface the reference model files use."""
    from tlxcv_amd.models.classification.resnet import BottleneckBlock
    from tlxcv_amd.tlx import nn
    down = nn.Sequential([nn.GroupConv2d(in_channels=32, out_channels=64, kernel_size=1, stride=2, b_init=(),
                                         padding=0, data_format="channels_first"),
                          nn.BatchNorm2d(num_features=64, data_format="channels_first")])
    blk = BottleneckBlock(32, 16, stride=2, downsample=down)
    blk.load_dict(seeded.fill(seeded.shapes_of(blk), 3))
    blk = blk.to(dev).set_eval()
    x = torch.from_numpy(seeded.image_batch(2, 1, hw=20, c=32)).to(dev)
    fused = blk(x)
    out = blk.relu(blk.bn1(blk.conv1(x)))
    out = blk.relu(blk.bn2(blk.conv2(out)))
    out = blk.bn3(blk.conv3(out))
    out = out + blk.downsample(x)
    out = blk.relu(out)
    torch.testing.assert_close(fused.contiguous(), out.contiguous(), atol=1e-5, rtol=1e-5)
