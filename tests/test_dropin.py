"""The reference's inference demo flow (demo/image_classification/predict.py:1-35, predict-vit.py) re-typed
against the aliased module names: `import tensorlayerx`, `tensorlayerx.vision.transforms`, `tlxcv.models`,
`tlxcv.tasks`.  The CPU part checks the host-side preprocessing; the gpu part runs the whole script body."""
import numpy as np
import pytest
import torch


def _png(tmp_path, hw=(180, 240)):
    from PIL import Image
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (hw[0], hw[1], 3), dtype=np.uint8)
    f = tmp_path / "dog.png"
    Image.fromarray(a).save(f)
    return str(f), a


def test_transform_pipeline_matches_numpy(tmp_path):
    import tlxcv_amd
    tlxcv_amd.install()
    from tensorlayerx.vision.transforms import Compose, Normalize, Resize, ToTensor
    from tensorlayerx.vision.transforms.utils import load_image
    f, a = _png(tmp_path)
    img = load_image(f)
    assert img.dtype == np.uint8 and np.array_equal(img, a)
    mean, std = (125.31, 122.95, 113.86), (62.99, 62.09, 66.70)
    t = Compose([Resize((224, 224)), Normalize(mean=mean, std=std), ToTensor(data_format="CHW")])(img)
    assert isinstance(t, torch.Tensor) and tuple(t.shape) == (3, 224, 224) and t.dtype == torch.float32
    from PIL import Image
    ref = (np.asarray(Image.fromarray(a).resize((224, 224), Image.BILINEAR)).astype(np.float32) - np.float32(mean)) / np.float32(std)
    np.testing.assert_allclose(t.cpu().numpy(), ref.transpose(2, 0, 1), rtol=0, atol=1e-6)
    # uint8 straight into ToTensor is rescaled to [0, 1]
    u = ToTensor(data_format="HWC")(img)
    assert float(u.max()) <= 1.0 and tuple(u.shape) == (180, 240, 3)


@pytest.mark.gpu
def test_predict_script_body_runs_on_the_engine(dev, tmp_path, fp32_mode):
    import tlxcv_amd
    tlxcv_amd.install()
    import tensorlayerx as tlx
    from tensorlayerx.vision.transforms import Compose, Normalize, Resize, ToTensor
    from tensorlayerx.vision.transforms.utils import load_image
    from tlxcv.models import resnet18, vit_small_patch16_224
    from tlxcv.tasks import ImageClassification
    from tlxcv_amd import seeded
    from oracle import functional as OF

    data_format, data_format_short = ('channels_last', 'HWC') if tlx.BACKEND == 'tensorflow' else ('channels_first', 'CHW')
    f, _ = _png(tmp_path)
    for ctor, oracle in ((lambda: resnet18(data_format=data_format, num_classes=10), lambda p, x: OF.resnet(p, x, 18)),
                         (lambda: vit_small_patch16_224(data_format=data_format, num_classes=10),
                          lambda p, x: OF.vit(p, x, "vit_small_patch16_224"))):
        backbone = ctor()
        model = ImageClassification(backbone)
        params = seeded.fill(seeded.shapes_of(backbone), 3)
        backbone.load_dict(params)
        w = str(tmp_path / "model.npz")
        model.save_weights(w)
        model.load_weights(w)                                   # predict.py:19  (positional TLX .npz)
        model.set_eval()                                        # :20  (lazy device placement: no extra line)
        assert next(model.parameters()).is_cuda
        image = load_image(f)                                   # :22
        transform = Compose([Resize((224, 224)), Normalize(mean=(125.31, 122.95, 113.86), std=(62.99, 62.09, 66.70)),
                             ToTensor(data_format=data_format_short)])
        image = tlx.expand_dims(transform(image), 0).cpu()      # :27-29; a HOST tensor, as a script on a CPU-default backend has
        class_id = tlx.convert_to_numpy(model.predict(image)).item()   # :31
        with torch.no_grad():
            ref = oracle({k: torch.from_numpy(v) for k, v in params.items()}, image.cpu())
        assert class_id == int(ref.argmax(-1))


def test_resample_tables_match_pil_bit_for_bit():
    """The integer resampler tables the device kernel consumes (tlx/vision/transforms/resample.py, restated from Pillow's
    published algorithm) against PIL itself, through a numpy run of the same two passes: up- and down-scaling, both axes."""
    from PIL import Image
    from tlxcv_amd.tlx.vision.transforms.resample import resize_u8_numpy
    rng = np.random.default_rng(1)
    for H, W, oh, ow in ((180, 240, 224, 224), (37, 53, 224, 224), (500, 375, 224, 224), (224, 300, 64, 96), (224, 224, 224, 224)):
        a = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        for interp, pil in (("bilinear", Image.BILINEAR), ("bicubic", Image.BICUBIC)):
            assert np.array_equal(resize_u8_numpy(a, (oh, ow), interp), np.asarray(Image.fromarray(a).resize((ow, oh), pil))), (H, W, interp)


@pytest.mark.gpu
def test_device_preprocessing_is_bit_identical_to_the_host_pipeline(dev, tmp_path):
    """Compose([Resize, Normalize, ToTensor]) as one device launch == the PIL / numpy host transforms,
    bit for bit, single image (the demo's call) and batched; and the space-to-depth output == the stem's own layout pass."""
    import tlxcv_amd
    from tlxcv_amd import engine as E
    from tlxcv_amd.tlx.vision.transforms import Compose, Normalize, Resize, ToTensor
    rng = np.random.default_rng(2)
    mean, std = (125.31, 122.95, 113.86), (62.99, 62.09, 66.70)
    for H, W in ((180, 240), (333, 500), (224, 224), (60, 41)):
        imgs = rng.integers(0, 256, (3, H, W, 3), dtype=np.uint8)
        for fmt in ("CHW", "HWC"):
            pipe = Compose([Resize((224, 224)), Normalize(mean=mean, std=std), ToTensor(data_format=fmt)])
            host = np.stack([_host(pipe, imgs[i]) for i in range(3)])
            one = pipe(imgs[0])                                       # the demo's call: numpy uint8 image in
            assert one.is_cuda and np.array_equal(one.cpu().numpy(), host[0])
            assert np.array_equal(pipe.batch(imgs).cpu().numpy(), host)
    # uint8 without Normalize: ToTensor's 1/255 rule
    pipe = Compose([Resize((96, 64), interpolation="bicubic"), ToTensor(data_format="CHW")])
    img = rng.integers(0, 256, (150, 90, 3), dtype=np.uint8)
    assert np.array_equal(pipe(img).cpu().numpy(), _host(pipe, img))
    # straight into the stem's space-to-depth layout (fp16): equals the layout kernel applied to the NCHW tensor
    pipe = Compose([Resize((224, 224)), Normalize(mean=mean, std=std), ToTensor(data_format="CHW")])
    imgs = rng.integers(0, 256, (2, 300, 260, 3), dtype=np.uint8)
    x = pipe.batch(imgs)
    assert torch.equal(pipe.batch(imgs, dtype=torch.float16, fold=2), E.nchw_to_nhwc_s2d(x, 2, torch.float16))


def _host(pipe, img):
    """The same Compose run through the host transforms one by one (PIL / numpy)."""
    data = img
    for t in pipe.transforms:
        data = t(data)
    return data.cpu().numpy() if isinstance(data, torch.Tensor) else np.asarray(data)
