"""YOLOv3 post-processing on the device: tlxmi_yolo_box against the restatement of Paddle's yolo_box
(oracle/detection.py; UNPINNED — the op is Paddle-only in the reference) and tlxmi_multiclass_nms against detections the
reference's own tlx_multiclass_nms produced (tests/golden/yolov3_post_b1.npz), plus the YOLOv3 model end to end."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import detection as OD
from tlxcv_amd import engine as E, seeded

pytestmark = pytest.mark.gpu


def _rows(det, cnt):
    det, cnt = det.cpu().numpy(), cnt.cpu().numpy()
    return np.concatenate([np.zeros((0, 6), np.float32)] + [det[i, :c] for i, c in enumerate(cnt)]), cnt


@pytest.mark.parametrize("tag", ["yolo", "dense"])
def test_multiclass_nms_reproduces_the_reference_function(dev, tag):
    g = np.load(os.path.join(GOLDEN, "yolov3_post_b1.npz"))
    b, s = torch.from_numpy(g[f"{tag}_boxes"]).to(dev), torch.from_numpy(g[f"{tag}_scores"]).to(dev)
    det, cnt = E.multiclass_nms(b, s, float(g[f"{tag}_thr"]), 0.5, 100)
    rows, cnt = _rows(det, cnt)
    assert cnt.tolist() == g[f"{tag}_counts"].tolist()
    assert np.array_equal(rows, g[f"{tag}_det"])                 # same boxes, same order, same bits (nothing is recomputed)


def test_multiclass_nms_edge_cases(dev):
    rng = np.random.default_rng(7)
    # nothing above the threshold; everything identical (one survivor per class); more survivors than keep_top_k; M not a power of two
    b = torch.from_numpy(rng.uniform(0, 50, (2, 37, 4)).astype(np.float32))
    b[..., 2:] += b[..., :2]
    s = torch.from_numpy(rng.random((2, 37, 3)).astype(np.float32))
    for thr, k in ((2.0, 10), (0.0, 5), (0.3, 100)):
        det, cnt = E.multiclass_nms(b.to(dev), s.to(dev), thr, 0.5, k)
        want = OD.multiclass_nms(b, s, thr, 0.5, k)
        rows, cnt = _rows(det, cnt)
        assert cnt.tolist() == [0 if w is None else w.shape[0] for w in want]
        assert np.array_equal(rows, np.concatenate([np.zeros((0, 6), np.float32)] + [w.numpy() for w in want if w is not None]))
    same = torch.tensor([[[0.0, 0.0, 10.0, 10.0]] * 6], dtype=torch.float32)
    sc = torch.tensor([[[0.9, 0.1], [0.8, 0.1], [0.1, 0.7], [0.1, 0.6], [0.5, 0.2], [0.2, 0.4]]], dtype=torch.float32)
    det, cnt = E.multiclass_nms(same.to(dev), sc.to(dev), 0.05, 0.5, 10)
    assert int(cnt[0]) == 2 and det[0, :2, 0].tolist() == [0.0, 1.0] and det[0, :2, 1].tolist() == pytest.approx([0.9, 0.7])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
def test_yolo_box_decode(dev, dtype):
    rng = np.random.default_rng(3)
    C, anchors = 5, [[116, 90, 156, 198, 373, 326], [30, 61, 62, 45, 59, 119]]
    heads = [torch.from_numpy(rng.standard_normal((2, 3 * (5 + C), h, w)).astype(np.float32)) for h, w in ((3, 5), (6, 10))]
    heads = [h.to(dtype).float() for h in heads]
    im_shape = torch.tensor([[96, 160], [90, 150]], dtype=torch.float32)
    want_b, want_s = OD.yolo_decode(heads, anchors, C, im_shape, torch.ones_like(im_shape), conf_thresh=0.3, downsample_ratio=32)
    nhwc = [h.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev) for h in heads]
    got_b, got_s = E.yolo_box(nhwc, anchors, C, im_shape.to(torch.int32), conf_thresh=0.3, downsample_ratio=32)
    assert got_b.shape == want_b.shape == (2, 3 * (15 + 60), 4)
    assert (want_b.abs().sum(-1) == 0).any() and (want_b.abs().sum(-1) > 0).any()        # both sides of conf_thresh occur
    torch.testing.assert_close(got_b.cpu(), want_b, atol=2e-3, rtol=1e-5)                  # exp / sigmoid: libm vs device, pixels
    torch.testing.assert_close(got_s.cpu(), want_s, atol=1e-6, rtol=1e-5)


def test_yolov3_forward_ends_in_detections(dev, fp32_mode):
    """YOLOv3.forward in eval mode (yolov3.py:51-104): head maps -> decode -> NMS -> labels / scores / boxes / bbox_num."""
    from tlxcv_amd import models
    g = np.load(os.path.join(GOLDEN, "yolov3_b1.npz"))
    m = models.YOLOv3()
    params = seeded.fill(seeded.shapes_of(m), int(g["weight_seed"]))
    m.load_dict(params)
    m = m.to(dev).set_eval()
    x = torch.from_numpy(seeded.image_batch(1, int(g["input_seed"]), hw=int(g["hw"])))
    out = m({"images": x.to(dev)})
    heads = [torch.from_numpy(g[f"head{i}"]) for i in range(3)]
    hw = int(g["hw"])
    im = torch.tensor([[hw, hw]], dtype=torch.float32)
    b, s = OD.yolo_decode(heads, m.yolo_head.mask_anchors, 92, im, torch.ones_like(im))
    want = OD.multiclass_nms(b, s, 0.01, 0.5, 100)[0]
    n = int(out["bbox_num"])
    assert n == (0 if want is None else want.shape[0]) and n > 0
    assert out["labels"].tolist() == want[:, 0].int().tolist()
    np.testing.assert_allclose(out["scores"], want[:, 1].numpy(), rtol=1e-4, atol=1e-5)
    assert np.abs(out["boxes"] - want[:, 2:].numpy().astype(int)).max() <= 1          # integer pixel boxes (cvt_results)
