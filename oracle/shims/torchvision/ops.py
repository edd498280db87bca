"""torchvision.ops.nms / batched_nms as documented: greedy suppression in descending score order, a box is dropped
when its IoU with an already kept box is > iou_threshold; returns kept indices sorted by score (ties: lower index
first).  batched_nms suppresses within a category only (torchvision's coordinate-offset trick)."""
import torch


def _iou_one_to_many(b, bs):
    lt = torch.maximum(b[:2], bs[:, :2])
    rb = torch.minimum(b[2:], bs[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    a = (b[2] - b[0]) * (b[3] - b[1])
    as_ = (bs[:, 2] - bs[:, 0]) * (bs[:, 3] - bs[:, 1])
    return inter / (a + as_ - inter)


def nms(boxes, scores, iou_threshold):
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    order = torch.sort(scores, descending=True, stable=True).indices
    boxes = boxes[order]
    n = boxes.shape[0]
    dead = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if dead[i]:
            continue
        keep.append(i)
        if i + 1 < n:
            dead[i + 1:] |= _iou_one_to_many(boxes[i], boxes[i + 1:]) > iou_threshold
    return order[torch.tensor(keep, dtype=torch.int64)]


def batched_nms(boxes, scores, idxs, iou_threshold):
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    off = idxs.to(boxes) * (boxes.max() + 1)
    return nms(boxes + off[:, None], scores, iou_threshold)
