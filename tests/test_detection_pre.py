"""The detection demo's input pipeline (demo/object_detection/transforms.py:96-246, predict-YOLOv3.py:54-61):
Resize(size=800, max_size=1333, auto_divide=32) -> Normalize, on the host classes (tlx/vision/transforms/detection.py), on the
device kernel (tlxmi_preprocess_linear_u8) and in the oracle's independent restatement of cv2.resize(INTER_LINEAR)
(oracle/detection.py; UNPINNED: OpenCV is not in this image)."""
import numpy as np
import pytest
import torch

from oracle import detection as OD
from tlxcv_amd.tlx.vision.transforms import detection as TD

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def test_output_size_follows_the_reference_arithmetic():
    # hand-computed from transforms.py:114-152 (size 800, max 1333, divisible by 32)
    assert TD.output_size((480, 640), 800, 1333, 32) == (800, 1088)       # 800 x 1066 -> 1066 rounded up to 1088
    assert TD.output_size((640, 480), 800, 1333, 32) == (1088, 800)
    assert TD.output_size((500, 2000), 800, 1333, 32) == (352, 1344)      # long side capped: shape = round(1333 * 500 / 2000) = 333
    assert TD.output_size((800, 800), 800, 1333, None) == (800, 800)      # already the target: unchanged
    assert TD.output_size((100, 50), (64, 32), None, None) == (32, 64)    # explicit (w, h) pair as cv2 takes it
    rng = np.random.default_rng(1)
    for _ in range(200):
        h, w = int(rng.integers(8, 3000)), int(rng.integers(8, 3000))
        size, mx = int(rng.integers(8, 1200)), int(rng.integers(8, 2000))
        ad = [None, 8, 32][int(rng.integers(0, 3))]
        ow, oh = OD.detection_resize_size((h, w), size, mx, ad)
        assert TD.output_size((h, w), size, mx, ad) == (oh, ow), (h, w, size, mx, ad)


@pytest.mark.parametrize("h,w,size,mx,ad", [(37, 53, 24, 40, 8), (60, 31, 32, 48, 32), (20, 90, 16, 40, 32), (9, 9, 64, 64, None),
                                            (64, 48, 16, 100, None)])
def test_host_resize_equals_the_oracle_restatement(h, w, size, mx, ad):
    rng = np.random.default_rng(h * w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    (out, lab) = TD.Compose([TD.Resize(size, mx, ad), TD.Normalize(MEAN, STD)])((img, None))
    want = OD.detection_preprocess(img, size, mx, ad, MEAN, STD)
    assert lab is not None and tuple(lab["im_shape"]) == (h, w) and tuple(lab["orig_size"]) == (w, h)
    out = out.cpu().numpy() if isinstance(out, torch.Tensor) else out
    assert out.dtype == np.float32 and out.shape == want.shape and np.array_equal(out, want)
    assert tuple(lab["size"]) == (want.shape[1], want.shape[0])


def test_labels_follow_the_image():
    img = np.zeros((50, 100, 3), dtype=np.uint8)
    lab = {"boxes": np.asarray([[10., 5., 30., 25.]], dtype=np.float32), "area": np.asarray([400.], dtype=np.float32)}
    rs = TD.Resize(100, 400, None)                       # 50 x 100 -> 100 x 200
    img2, lab2 = rs((img, lab))
    assert img2.shape == (100, 200, 3)
    assert np.allclose(lab2["boxes"], [[20., 10., 60., 50.]]) and np.allclose(lab2["area"], [1600.])
    assert tuple(lab2["size"]) == (200, 100) and np.allclose(lab2["scale_factor"], [2.0, 2.0])
    _, lab3 = TD.Normalize(MEAN, STD)((img2, lab2))
    assert np.allclose(lab3["boxes"], [[0.2, 0.3, 0.2, 0.4]])       # centre format, relative to (w, h)
    with pytest.raises(NotImplementedError):
        rs((img, {"masks": np.zeros((1, 50, 100))}))


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["HWC", "CHW"])
@pytest.mark.parametrize("h,w,size,mx,ad", [(37, 53, 24, 40, 8), (480, 640, 800, 1333, 32), (427, 640, 800, 1333, 32), (500, 353, 320, 512, 32)])
def test_device_pipeline_is_bit_identical_to_the_host_arithmetic(dev, h, w, size, mx, ad, layout):
    rng = np.random.default_rng(h + w)
    imgs = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
    oh, ow = TD.output_size((h, w), size, mx, ad)
    got = TD.device_resize_normalize(torch.from_numpy(imgs).to(dev), (oh, ow), MEAN, STD, layout=layout).cpu().numpy()
    for n in range(2):
        want = (TD._resize_host(imgs[n], oh, ow).astype(np.float32) / 255.0 - np.asarray(MEAN, np.float32)) / np.asarray(STD, np.float32)
        g = got[n] if layout == "HWC" else np.transpose(got[n], (1, 2, 0))
        assert np.array_equal(g, want), (n, np.abs(g - want).max())
    if h < 100:                                                           # the oracle's per-pixel loop: small cases only
        want = OD.detection_preprocess(imgs[0], size, mx, ad, MEAN, STD)
        g = got[0] if layout == "HWC" else np.transpose(got[0], (1, 2, 0))
        assert np.array_equal(g, want)


@pytest.mark.gpu
def test_compose_runs_on_the_device_and_matches_the_host_classes(dev):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (120, 200, 3), dtype=np.uint8)
    tr = TD.Compose([TD.Resize(size=160, max_size=256, auto_divide=32), TD.Normalize(MEAN, STD)])
    out, lab = tr((img, None))
    assert isinstance(out, torch.Tensor) and out.is_cuda and tuple(out.shape) == (160, 256, 3)      # 160 x 266 capped -> 154 x 256 -> /32
    host_img, host_lab = TD.Normalize(MEAN, STD)(TD.Resize(160, 256, 32)((img, None)))
    assert np.array_equal(out.cpu().numpy(), host_img)
    assert all(np.array_equal(lab[k], host_lab[k]) for k in host_lab)
    # fp16 output = the fp32 result rounded once; 1-channel images; without Normalize the value is v / 255
    g16 = TD.device_resize_normalize(torch.from_numpy(img).to(dev)[None], (160, 256), MEAN, STD, dtype=torch.float16)[0]
    assert torch.equal(g16.cpu(), torch.from_numpy(host_img).half())
    gray = rng.integers(0, 256, (33, 47, 1), dtype=np.uint8)
    g1 = TD.device_resize_normalize(torch.from_numpy(gray).to(dev)[None], (64, 96))[0].cpu().numpy()
    assert np.array_equal(g1, TD._resize_host(gray, 64, 96).astype(np.float32) / 255.0)
