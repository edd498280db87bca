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


def test_multiclass_nms_returns_the_kept_boxes_indices(dev):
    """The index-returning form (yolov3.py:70-78 `nms_keep_idx`): same detections as the plain call, each row's index points at its
    box, -1 past the count; the golden candidates of the reference's own function, a ragged random case with ties, and the
    one-thread-per-row (keep_top_k > 1024) form of the kernel."""
    g = np.load(os.path.join(GOLDEN, "yolov3_post_b1.npz"))
    cases = [(torch.from_numpy(g[f"{t}_boxes"]), torch.from_numpy(g[f"{t}_scores"]), float(g[f"{t}_thr"]), 100) for t in ("yolo", "dense")]
    rng = np.random.default_rng(11)
    b = torch.from_numpy(rng.uniform(0, 60, (3, 301, 4)).astype(np.float32))
    b[..., 2:] += b[..., :2]
    s = torch.from_numpy((rng.integers(0, 50, (3, 301, 4)) / 50.0).astype(np.float32))      # many equal scores
    cases += [(b, s, 0.3, 20), (b, s, 0.5, 1500), (b[:, :37], s[:, :37], 2.0, 10)]
    for bx, sc, thr, k in cases:
        det0, cnt0 = E.multiclass_nms(bx.to(dev), sc.to(dev), thr, 0.5, k)
        det, cnt, idx = E.multiclass_nms(bx.to(dev), sc.to(dev), thr, 0.5, k, return_index=True)
        assert torch.equal(det, det0) and torch.equal(cnt, cnt0) and idx.shape == (bx.shape[0], k) and idx.dtype == torch.int32
        want = OD.multiclass_nms(bx, sc, thr, 0.5, k, return_index=True)
        for n, w in enumerate(want):
            c = int(cnt[n])
            assert c == (0 if w is None else w[0].shape[0])
            assert (idx[n, c:] == -1).all()
            if c:
                assert torch.equal(bx[n][idx[n, :c].cpu().long()], det[n, :c, 2:].cpu())          # the row's box IS that box
                assert torch.equal(sc[n][idx[n, :c].cpu().long()].max(1).values, det[n, :c, 1].cpu())
                assert np.array_equal(det[n, :c].cpu().numpy(), w[0].numpy())
                # ties in score may be kept in either order by a stable / unstable sort only if rows are identical: compare as sets per score
                assert sorted(idx[n, :c].tolist()) == sorted(w[1].tolist())


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
    # for_mot (:64-78): the same detections, the neck's embedding maps, and which candidate box each row came from
    m.for_mot = True
    mot = m({"images": x.to(dev)})
    assert torch.equal(mot["detections"], out["detections"]) and len(mot["emb_feats"]) == 3
    keep = mot["nms_keep_idx"]
    assert keep.shape == (1, 100) and (keep[0, n:] == -1).all() and int(keep[0, :n].min()) >= 0 and keep[0, :n].unique().numel() == n


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
def test_iou_aware_objectness(dev, dtype):
    """tlxmi_yolo_iou_aware against the reference lines restated (yolov3.py:355-376): IoU channels in front, dropped on output."""
    rng = np.random.default_rng(31)
    A, C, H, W = 3, 7, 5, 6
    x = torch.from_numpy(rng.standard_normal((2, A * (6 + C), H, W)).astype(np.float32) * 2.0)
    x[0, 0, 0, 0], x[0, A + 4, 0, 0] = 40.0, 40.0                    # saturated sigmoid: the clip of _de_sigmoid decides
    x[0, 1, 0, 1], x[0, A + (5 + C) + 4, 0, 1] = -40.0, -40.0
    if dtype == torch.float16:
        x = x.half().float()
    want = OD.yolo_iou_aware(x, A, 0.4)
    got = E.yolo_iou_aware(x.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev), A, C, 0.4)
    assert got.shape == (2, H, W, A * (5 + C))
    tol = dict(atol=2e-5, rtol=2e-5) if dtype == torch.float32 else dict(atol=1e-2, rtol=2e-3)
    torch.testing.assert_close(got.float().cpu().permute(0, 3, 1, 2), want, **tol)


def test_yolov3_head_iou_aware_and_neck_for_mot(dev, fp32_mode):
    from tlxcv_amd import models
    from oracle import functional as OF
    rng = np.random.default_rng(32)
    head = models.YOLOv3Head(in_channels=[64, 32], anchors=[[10, 13], [16, 30], [33, 23], [30, 61]], anchor_masks=[[2, 3], [0, 1]],
                             num_classes=5, iou_aware=True, iou_aware_factor=0.3)
    params = seeded.fill(seeded.shapes_of(head), 9)
    head.load_dict(params)
    head = head.to(dev).set_eval()
    feats = [torch.from_numpy(rng.standard_normal((1, 64, 4, 4)).astype(np.float32)), torch.from_numpy(rng.standard_normal((1, 32, 8, 8)).astype(np.float32))]
    outs = head([f.to(dev) for f in feats])
    assert [tuple(o.shape) for o in outs] == [(1, 2 * 10, 4, 4), (1, 2 * 10, 8, 8)]
    for i, (f, o) in enumerate(zip(feats, outs)):
        w, b = torch.from_numpy(params[f"yolo_outputs_{i}.filters"]), torch.from_numpy(params[f"yolo_outputs_{i}.biases"])
        want = OD.yolo_iou_aware(torch.nn.functional.conv2d(f, w, b), 2, 0.3)
        torch.testing.assert_close(o.cpu(), want, atol=1e-4, rtol=1e-4)
    # neck with for_mot (yolov3.py:243-257): same tips, plus one route map per level
    fpn = models.YOLOv3FPN(in_channels=[64, 128, 256])
    fpn.load_dict(seeded.fill(seeded.shapes_of(fpn), 10))
    fpn = fpn.to(dev).set_eval()
    body = [torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(dev) for s in ((1, 64, 16, 16), (1, 128, 8, 8), (1, 256, 4, 4))]
    plain = fpn(body)
    mot = fpn(body, for_mot=True)
    assert set(mot) == {"yolo_feats", "emb_feats"} and len(mot["emb_feats"]) == 3
    assert all(torch.equal(a, b) for a, b in zip(plain, mot["yolo_feats"]))
    assert [tuple(t.shape[1:]) for t in mot["emb_feats"]] == [(512, 4, 4), (256, 8, 8), (128, 16, 16)]      # the blocks' route maps (YoloDetBlock width 512 / 256 / 128)
