"""CPU-side tests of the host layer: module trees mirror the reference, fixtures load, and the
product refuses to compute on CPU tensors (there is no fallback path to hide behind)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


def test_resnet50_parameter_tree_matches_reference_fixture():
    from tlxcv_amd.models import resnet50
    from tlxcv_amd import seeded
    names = list(np.load(os.path.join(GOLDEN, "resnet50_b4.npz"))["param_names"])
    m = resnet50()
    assert list(seeded.shapes_of(m).keys()) == names
    assert sum(p.numel() for p in m.parameters()) == 25557032


def test_vit_parameter_tree_matches_reference_fixture():
    from tlxcv_amd.models import vit_base_patch16_224
    from tlxcv_amd import seeded
    names = list(np.load(os.path.join(GOLDEN, "vit_b16_b2.npz"))["param_names"])
    assert list(seeded.shapes_of(vit_base_patch16_224()).keys()) == names


@pytest.mark.parametrize("fname,ctor", [("mobilenetv1_b2.npz", "MobileNetV1"), ("darknet53_b1.npz", "DarkNet"),
                                        ("yolov3_b1.npz", "YOLOv3"),
                                        ("swin_b_b2.npz", "swintransformer_base_patch4_window7_224")])
def test_other_parameter_trees_match_fixtures(fname, ctor):
    from tlxcv_amd import models, seeded
    names = list(np.load(os.path.join(GOLDEN, fname))["param_names"])
    assert list(seeded.shapes_of(getattr(models, ctor)()).keys()) == names


def test_cpu_tensor_is_refused():
    from tlxcv_amd.models import resnet18
    m = resnet18()
    m.set_eval()
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(1, 3, 32, 32))


def test_training_forward_is_refused():
    from tlxcv_amd.tlx import nn
    bn = nn.BatchNorm2d(num_features=8, data_format="channels_first")
    with pytest.raises(NotImplementedError, match="set_eval"):
        bn(torch.zeros(1, 8, 4, 4))


def test_sequential_accepts_list_and_varargs_and_lists_are_adopted():
    from tlxcv_amd.tlx import nn
    a = nn.Sequential([nn.ReLU(), nn.ReLU6()])
    b = nn.Sequential(nn.ReLU(), nn.ReLU6())
    assert len(a) == len(b) == 2

    class Holder(nn.Module):
        def __init__(self):
            super().__init__()
            self.blocks = []
            self.blocks.append(nn.Linear(in_features=4, out_features=4))

    h = Holder()
    h.set_eval()
    assert "blocks_0.weights" in h.state_dict() and h.blocks[0].is_train is False


def test_b_init_falsy_means_no_bias():
    from tlxcv_amd.tlx import nn
    for falsy in ((), False, None):
        assert nn.Linear(in_features=4, out_features=4, b_init=falsy).biases is None
        assert nn.GroupConv2d(in_channels=4, out_channels=4, kernel_size=1, b_init=falsy).biases is None
    assert nn.Linear(in_features=4, out_features=4).biases is not None


def test_install_aliases_tensorlayerx_and_tlxcv():
    import tlxcv_amd
    tlxcv_amd.install()
    import tensorlayerx as tlx
    import tensorlayerx.nn as nn
    from tlxcv.models import resnet50
    from tlxcv.tasks import ImageClassification
    assert tlx.BACKEND == "torch" and nn.GroupConv2d is tlxcv_amd.tlx.nn.GroupConv2d
    model = ImageClassification(resnet50(num_classes=10))
    assert model.backbone.fc.out_features == 10


def test_save_and_load_weights_roundtrip(tmp_path):
    from tlxcv_amd.models import resnet18
    from tlxcv_amd import seeded
    m = resnet18(num_classes=7)
    m.load_dict(seeded.fill(seeded.shapes_of(m), 5))
    f = str(tmp_path / "model.npz")
    m.save_weights(f)
    m2 = resnet18(num_classes=7)
    m2.load_weights(f)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
