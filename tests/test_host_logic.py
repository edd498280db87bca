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


# ---------------------------------------------------------------------------------------------
# TensorLayerX `.npz` checkpoint interchange (positional `params` list)
# ---------------------------------------------------------------------------------------------
def _tlx_npz_case(tag, tmp_path):
    """The positional checkpoint the reference's own model class wrote through the oracle stand-in's save_weights
    (oracle/gen_golden.py: gen_tlx_npz), re-materialised as the file a user would hand to load_weights."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "tlx_npz_small.npz"))
    n = int(g[f"{tag}_n"])
    params = np.empty(n, dtype=object)
    for i in range(n):
        params[i] = g[f"{tag}_params_{i:03d}"]
    path = str(tmp_path / f"{tag}_model.npz")
    np.savez(path, params=params)
    return g, path


def _small_models():
    from tlxcv_amd import models
    return {"vit": lambda: models.VisionTransformer(img_size=32, patch_size=8, num_classes=10, embed_dim=32, depth=2, num_heads=2,
                                                    mlp_ratio=2, qkv_bias=True, epsilon=1e-6),
            "mbv1": lambda: models.MobileNetV1(scale=0.125, num_classes=10)}


@pytest.mark.parametrize("tag", ["vit", "mbv1"])
def test_positional_tlx_npz_restores_every_named_weight(tag, tmp_path):
    from tlxcv_amd import seeded
    g, path = _tlx_npz_case(tag, tmp_path)
    m = _small_models()[tag]()
    m.load_weights(path)                                            # demo/image_classification/predict.py:19
    want = seeded.fill(seeded.shapes_of(m), int(g[f"{tag}_weight_seed"]))
    assert list(want) == [str(k) for k in g[f"{tag}_param_names"]]      # same parameter tree as the reference's class
    sd = m.state_dict()
    for k, v in want.items():
        assert np.array_equal(sd[k].numpy(), v), k
    # and back: save_weights('x.npz') writes the same positional list, array for array
    out = str(tmp_path / "again.npz")
    m.save_weights(out)
    a, b = np.load(out, allow_pickle=True)["params"], np.load(path, allow_pickle=True)["params"]
    assert len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))
    # the name-keyed form still loads, and derived buffers are never demanded of it
    m.save_weights(str(tmp_path / "dict.npz"), format="npz_dict")
    m2 = _small_models()[tag]()
    m2.load_weights(str(tmp_path / "dict.npz"))
    assert all(torch.equal(m2.state_dict()[k], sd[k]) for k in sd)


def test_positional_tlx_npz_rejects_a_foreign_model(tmp_path):
    _, path = _tlx_npz_case("vit", tmp_path)
    with pytest.raises(ValueError):
        _small_models()["mbv1"]().load_weights(path)


def test_a_name_keyed_checkpoint_is_never_unpickled(tmp_path):
    """only the single-key positional `params` form may run pickle; a name-keyed file that smuggles an object
    array is refused by numpy (allow_pickle=False) instead of being executed."""
    m = _small_models()["vit"]()
    sd = {k: v.numpy() for k, v in m.state_dict().items()}
    evil = np.empty(1, dtype=object)
    evil[0] = {"not": "an array"}
    path = str(tmp_path / "dict_with_object.npz")
    np.savez(path, **sd, extra=evil)
    m.load_weights(path)                     # plain arrays load; the object entry is never touched
    first = next(iter(sd))
    sd2 = dict(sd)
    sd2[first] = evil
    path2 = str(tmp_path / "dict_with_object_weight.npz")
    np.savez(path2, **sd2)
    with pytest.raises(ValueError, match="[Oo]bject arrays|allow_pickle"):
        m.load_weights(path2)


def test_compose_keeps_rgba_images_on_the_host_path(monkeypatch):
    """Pillow premultiplies alpha around a resize of an RGBA image, the device kernel does not: 4-channel images
    must take the host transforms (here: the device batch() must not be called even when a GPU is reported)."""
    from tlxcv_amd.tlx.vision import transforms as T
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    c = T.Compose([T.Resize((8, 8)), T.Normalize(mean=(0.5,), std=(0.5,)), T.HWC2CHW()])   # not a device plan at all
    c2 = T.Compose([T.Resize((8, 8)), T.ToTensor("CHW")])
    called = []
    monkeypatch.setattr(T.Compose, "batch", lambda self, *a, **k: called.append(1))
    rgba = np.random.default_rng(0).integers(0, 255, (16, 16, 4), dtype=np.uint8)
    try:
        c2(rgba)
    except Exception:
        pass                                  # ToTensor may try to reach the (absent) device: irrelevant here
    assert not called
    out = c(rgba)
    assert out.shape == (4, 8, 8) and not called


def test_derived_tensors_follow_parameter_updates():
    """a cached derived tensor must not survive load_state_dict / in-place parameter writes."""
    from tlxcv_amd.tlx import nn
    lin = nn.Linear(in_features=8, out_features=8)
    calls = []
    build = lambda: calls.append(1) or len(calls)      # noqa: E731
    assert lin._cached("k", build) == 1 and lin._cached("k", build) == 1
    with torch.no_grad():
        lin.weights.add_(1.0)                          # in-place edit: version counter moves
    assert lin._cached("k", build) == 2
    lin.load_state_dict(lin.state_dict())              # torch's own loader
    assert lin._cached("k", build) == 3
    bn = nn.BatchNorm2d(num_features=8)
    conv = nn.GroupConv2d(in_channels=8, out_channels=8, kernel_size=1, padding=0, b_init=None)
    assert conv._cached("f", build, deps=(bn,)) == 4 and conv._cached("f", build, deps=(bn,)) == 4
    with torch.no_grad():
        bn.moving_var.mul_(2.0)                        # the BatchNorm folded into the conv changed
    assert conv._cached("f", build, deps=(bn,)) == 5
    e0 = conv._weights_epoch
    conv.load_dict({"filters": torch.zeros(8, 8, 1, 1)}, strict=False)
    assert conv._weights_epoch == e0 + 1               # what a captured hipGraph compares before replay


def test_linear_tail_plan_matches_the_round_arithmetic():
    """Host logic of engine._linear_tail (no GPU): ViT-B/16 proj at batch 256 on 256 CUs -> 170 row tiles on the persistent
    kernel, 6912 rows on K slices; whole rounds, a last round over half full, too short a K -> no split."""
    from tlxcv_amd import engine
    import types
    class PK:      # noqa: E306
        def __init__(self, K, Cout):
            self.Cin = self.Cin_pad = K
            self.Cout = Cout
    class X:       # noqa: E306
        device = types.SimpleNamespace(index=0)
        dtype = torch.float16
        def element_size(self):
            return 2
    engine._cus[0] = 256
    assert engine._linear_tail(50432, 768, PK(768, 768), X()) is None                 # off by default (measured slower)
    engine.set_option("tail_splitk", True)
    try:
        assert engine._linear_tail(50432, 768, PK(768, 768), X())[0] == 170 * 256
        assert engine._linear_tail(50432, 3072, PK(3072, 768), X()) == (170 * 256, 2)
        assert engine._linear_tail(220 * 197, 768, PK(768, 768), X()) is None          # 170 x 3 = 510 tiles: whole rounds... of 256? 1.99 -> over half
        assert engine._linear_tail(50432, 768, PK(768, 2304), X()) is None            # 1773 tiles = 6 rounds + 237: over half full
        assert engine._linear_tail(50432, 256, PK(256, 768), X()) is None             # 4 K tiles: too short for slices
        with engine.shared_plan("half"):
            assert engine._linear_tail(25216, 768, PK(768, 768), X())[0] == 85 * 256
    finally:
        engine.set_option("tail_splitk", False)
        engine._cus.pop(0, None)
