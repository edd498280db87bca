"""CPU restatements of the YOLOv3 post-processing — TEST INFRASTRUCTURE (see oracle/__init__.py).

yolo_box:        the published algorithm of paddle.vision.ops.yolo_box (the op YOLOBox.__call__ calls through
                 `yolo_box_func`, tlxcv/models/detection/yolov3.py:558-579; the reference has it on the Paddle backend
                 only, detection/utils/ops.py:436-452).  UNPINNED: Paddle is not in this image and the reference holds no
                 vector for it; restated from the operator's documentation.
multiclass_nms:  tlx_multiclass_nms, tlxcv/models/detection/utils/ops.py:255-329, restated; oracle/gen_golden.py checks
                 this function against the reference's OWN function (loaded by path, torchvision.ops from oracle/shims).
"""
import torch


def yolo_box(x, img_size, anchors, class_num, conf_thresh, downsample_ratio, clip_bbox=True, scale_x_y=1.0):
    """x (N, A*(5+C), H, W) fp32, img_size (N, 2) int (h, w), anchors flat [w0, h0, w1, h1, ...] -> boxes (N, A*H*W, 4),
    scores (N, A*H*W, C)."""
    N, _, H, W = x.shape
    A, C = len(anchors) // 2, class_num
    x = x.reshape(N, A, 5 + C, H, W).float()
    img_h = img_size[:, 0].float().view(N, 1, 1, 1)
    img_w = img_size[:, 1].float().view(N, 1, 1, 1)
    gx = torch.arange(W, dtype=torch.float32).view(1, 1, 1, W)
    gy = torch.arange(H, dtype=torch.float32).view(1, 1, H, 1)
    aw = torch.tensor(anchors[0::2], dtype=torch.float32).view(1, A, 1, 1)
    ah = torch.tensor(anchors[1::2], dtype=torch.float32).view(1, A, 1, 1)
    bias = -0.5 * (scale_x_y - 1.0)
    conf = torch.sigmoid(x[:, :, 4])
    cx = (gx + torch.sigmoid(x[:, :, 0]) * scale_x_y + bias) * img_w / W
    cy = (gy + torch.sigmoid(x[:, :, 1]) * scale_x_y + bias) * img_h / H
    bw = torch.exp(x[:, :, 2]) * aw * img_w / (downsample_ratio * W)
    bh = torch.exp(x[:, :, 3]) * ah * img_h / (downsample_ratio * H)
    x1, y1, x2, y2 = cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2
    if clip_bbox:
        x1, y1 = x1.clamp(min=0), y1.clamp(min=0)
        x2, y2 = torch.minimum(x2, img_w - 1), torch.minimum(y2, img_h - 1)
    keep = (conf >= conf_thresh).float()
    boxes = torch.stack([x1, y1, x2, y2], -1) * keep.unsqueeze(-1)
    scores = (conf.unsqueeze(2) * torch.sigmoid(x[:, :, 5:])) * keep.unsqueeze(2)            # (N, A, C, H, W)
    return boxes.reshape(N, A * H * W, 4), scores.permute(0, 1, 3, 4, 2).reshape(N, A * H * W, C)


def yolo_decode(heads, mask_anchors, class_num, im_shape, scale_factor, conf_thresh=0.005, downsample_ratio=32, clip_bbox=True,
                scale_x_y=1.0):
    """YOLOBox.__call__, yolov3.py:558-579, with scores kept as (N, boxes, classes)."""
    origin = (im_shape / scale_factor).to(torch.int32)
    bl, sl = [], []
    for i, (h, anc) in enumerate(zip(heads, mask_anchors)):
        b, s = yolo_box(h, origin, anc, class_num, conf_thresh, downsample_ratio // 2 ** i, clip_bbox, scale_x_y)
        bl.append(b)
        sl.append(s)
    return torch.cat(bl, 1), torch.cat(sl, 1)


def _nms(boxes, scores, thr):
    order = torch.sort(scores, descending=True, stable=True).indices
    b = boxes[order]
    n = b.shape[0]
    dead = torch.zeros(n, dtype=torch.bool)
    keep = []
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    for i in range(n):
        if dead[i]:
            continue
        keep.append(i)
        if i + 1 < n:
            lt = torch.maximum(b[i, :2], b[i + 1:, :2])
            rb = torch.minimum(b[i, 2:], b[i + 1:, 2:])
            wh = (rb - lt).clamp(min=0)
            inter = wh[:, 0] * wh[:, 1]
            dead[i + 1:] |= inter / (area[i] + area[i + 1:] - inter) > thr
    return order[torch.tensor(keep, dtype=torch.int64)]


def multiclass_nms(bboxes, scores, score_threshold=0.7, nms_threshold=0.45, keep_top_k=100):
    """tlx_multiclass_nms, utils/ops.py:255-329 (class-aware branch): per image rows (class, score, x1, y1, x2, y2), None when
    nothing passes the threshold.  torchvision.ops.batched_nms = greedy NMS on boxes shifted by class * (max coordinate + 1)."""
    out = []
    for xyxy, score in zip(bboxes, scores):
        conf, pred = score.max(1)                                                           # :287-288 (first maximal class)
        m = conf >= score_threshold
        det = torch.cat([xyxy, conf[:, None], pred[:, None].float()], 1)[m]                 # :292-296
        if det.shape[0] == 0:
            out.append(None)
            continue
        off = det[:, 5] * (det[:, :4].max() + 1)
        keep = _nms(det[:, :4] + off[:, None], det[:, 4], nms_threshold)                    # :306-309
        det = det[keep]
        order = torch.argsort(det[:, 4], descending=True, stable=True)                      # :314-317
        if keep_top_k > 0 and len(order) > keep_top_k:
            order = order[:keep_top_k]
        det = det[order]
        out.append(torch.cat([det[:, 5:6], det[:, 4:5], det[:, :4]], 1))                    # :320-322
    return out
