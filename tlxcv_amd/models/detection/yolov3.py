"""YOLOv3 = DarkNet-53 -> FPN neck -> 1x1 head convs, forward only, on the MI355X engine.

Mirrors tlxcv/models/detection/yolov3.py: YoloDetBlock :122-183, YOLOv3FPN :186-258, YOLOv3Head :261-378
(head conv :311-322, 3*(num_classes+5) channels, bias).  In the neck, `interpolate(route, scale_factor=2)` +
`tlx.concat([route, x])` (:246-256) are two copy kernels into one pre-sized NHWC buffer (nearest x2 written
at channel offset 0, the backbone map at its channel offset) instead of materialising both.
Post-processing runs on the device too: YOLOBox (:541-579) = tlxmi_yolo_box — the published
algorithm of paddle.vision.ops.yolo_box, the op the reference calls and has on the Paddle backend only
(utils/ops.py:436-452) — and MultiClassNMS = tlxmi_multiclass_nms, following the reference's torch-side
tlx_multiclass_nms (utils/ops.py:255-329) with scores as (N, boxes, classes).  Off Paddle the reference's own glue between
the two is inconsistent (scores handed over transposed, a list where three values are unpacked: post_process.py:46), so
the end-to-end detections cannot be pinned against it; each stage is checked against its restatement (oracle/detection.py).
Out of scope: YOLOv3Loss, Gt2YoloTarget (training)."""
import numpy as np
import torch

from ... import engine as E
from ...tlx import nn
from ...tlx.nn import as_nhwc, from_nhwc
from .backbones.darknet import ConvBNLayer, DarkNet
from .backbones.mobilenet_v1 import MobileNet

__all__ = ["YOLOv3", "YOLOv3FPN", "YOLOv3Head", "YoloDetBlock"]


def create(obj, **kwds):
    if isinstance(obj, str):
        return {"DarkNet": DarkNet, "MobileNet": MobileNet}[obj](**kwds)      # the reference evals the name (yolov3.py:16-20)
    return obj


class YoloDetBlock(nn.Module):
    def __init__(self, ch_in, channel, norm_type="bn", freeze_norm=False, name="", data_format="channels_first"):
        super().__init__()
        self.ch_in, self.channel = ch_in, channel
        assert channel % 2 == 0, "channel {} cannot be divided by 2".format(channel)
        conv_def = [["conv0", ch_in, channel, 1, ".0.0"], ["conv1", channel, channel * 2, 3, ".0.1"],
                    ["conv2", channel * 2, channel, 1, ".1.0"], ["conv3", channel, channel * 2, 3, ".1.1"],
                    ["route", channel * 2, channel, 1, ".2"]]
        self.conv_module = nn.Sequential([
            ConvBNLayer(ch_in=ci, ch_out=co, filter_size=fs, padding=(fs - 1) // 2, data_format=data_format,
                        name=name + cn + pn) for cn, ci, co, fs, pn in conv_def])
        self.tip = ConvBNLayer(ch_in=channel, ch_out=channel * 2, filter_size=3, padding=1, data_format=data_format,
                               name=name + ".tip")

    def run_nhwc(self, v):
        for l in self.conv_module:
            v = l.run_nhwc(v)
        return v, self.tip.run_nhwc(v)

    def forward(self, inputs):
        route = self.conv_module(inputs)
        return route, self.tip(route)


class YOLOv3FPN(nn.Module):
    def __init__(self, in_channels=[256, 512, 1024], norm_type="bn", freeze_norm=False, data_format="channels_first"):
        super().__init__()
        assert len(in_channels) > 0, "in_channels length should > 0"
        self.in_channels, self.num_blocks, self.data_format = in_channels, len(in_channels), data_format
        self._out_channels = []
        self.yolo_blocks, self.routes = [], []
        for i, in_channel in enumerate(in_channels[::-1]):
            if i > 0:
                in_channel += 512 // 2 ** i
            self.yolo_blocks.append(YoloDetBlock(in_channel, channel=512 // 2 ** i, data_format=data_format,
                                                 name="yolo_block.{}".format(i)))
            self._out_channels.append(1024 // 2 ** i)
            if i < self.num_blocks - 1:
                self.routes.append(ConvBNLayer(ch_in=512 // 2 ** i, ch_out=256 // 2 ** i, filter_size=1, stride=1,
                                               padding=0, data_format=data_format, name="yolo_transition.{}".format(i)))

    def _run(self, feats, emb=None):
        assert len(feats) == self.num_blocks
        feats = feats[::-1]
        out, route = [], None
        for i, x in enumerate(feats):
            if i > 0:                                   # concat([upsample2x(route), x]) along channels
                N, H, W, Cx = x.shape
                Cr = route.shape[-1]
                cat = torch.empty((N, H, W, Cr + Cx), dtype=x.dtype, device=x.device)
                E.upsample2x_into(route, cat, 0)
                E.copy_channels_into(x, cat, Cr)
                x = cat
            route, tip = self.yolo_blocks[i].run_nhwc(x)
            out.append(tip)
            if emb is not None:
                emb.append(route)                       # :250-251, the block's route BEFORE the transition conv
            if i < self.num_blocks - 1:
                route = self.routes[i].run_nhwc(route)
        return out

    def run_nhwc(self, feats, emb=None):
        return self._run(feats, emb)

    def forward(self, X, for_mot=False):
        """for_mot (:243-257): also the per-level route maps, as {"yolo_feats": [...], "emb_feats": [...]}."""
        emb = [] if for_mot else None
        tips = [from_nhwc(t, self.data_format) for t in self._run([as_nhwc(x, self.data_format) for x in X], emb)]
        if for_mot:
            return {"yolo_feats": tips, "emb_feats": [from_nhwc(t, self.data_format) for t in emb]}
        return tips


class YOLOv3Head(nn.Module):
    def __init__(self, in_channels=[1024, 512, 256],
                 anchors=[[10, 13], [16, 30], [33, 23], [30, 61], [62, 45], [59, 119], [116, 90], [156, 198], [373, 326]],
                 anchor_masks=[[6, 7, 8], [3, 4, 5], [0, 1, 2]], num_classes=92, loss=None, batch_transforms=None,
                 iou_aware=False, iou_aware_factor=0.4, data_format="channels_first"):
        super().__init__()
        assert len(in_channels) > 0, "in_channels length should > 0"
        self.iou_aware, self.iou_aware_factor = bool(iou_aware), float(iou_aware_factor)
        self.in_channels, self.num_classes, self.data_format = in_channels, num_classes, data_format
        self.parse_anchor(anchors, anchor_masks)
        self.num_outputs = len(self.anchors)
        self.yolo_outputs = []
        for i, anc in enumerate(self.anchors):
            self.yolo_outputs.append(nn.GroupConv2d(
                in_channels=self.in_channels[i], out_channels=len(anc) * (self.num_classes + (6 if self.iou_aware else 5)), kernel_size=1, stride=1,
                padding=0, data_format=data_format, b_init=nn.initializers.xavier_uniform(),
                W_init=nn.initializers.HeNormal(), name="yolo_output.{}".format(i)))

    def parse_anchor(self, anchors, anchor_masks):
        self.anchors = [[anchors[i] for i in mask] for mask in anchor_masks]
        self.mask_anchors = []
        for masks in anchor_masks:
            self.mask_anchors.append([])
            for mask in masks:
                assert mask < len(anchors), "anchor mask index overflow"
                self.mask_anchors[-1].extend(anchors[mask])

    def run_nhwc(self, feats):
        assert len(feats) == len(self.anchors)
        outs = [fn.run_nhwc(f) for fn, f in zip(self.yolo_outputs, feats)]
        if self.iou_aware:                              # :355-376: IoU-weighted objectness, the IoU channels dropped
            outs = [E.yolo_iou_aware(o.contiguous(), len(anc), self.num_classes, self.iou_aware_factor) for o, anc in zip(outs, self.anchors)]
        return outs

    def forward(self, outputs, targets=None):
        if targets is not None:
            raise NotImplementedError("YOLOv3Loss is out of scope (training)")
        feats = outputs["neck_feats"] if isinstance(outputs, dict) else outputs
        return [from_nhwc(y, self.data_format) for y in self.run_nhwc([as_nhwc(f, self.data_format) for f in feats])]


class YOLOBox:
    """yolov3.py:541-579; __call__ takes the head maps as NHWC engine tensors."""

    def __init__(self, num_classes=92, conf_thresh=0.005, downsample_ratio=32, clip_bbox=True, scale_x_y=1.0,
                 data_format="channels_first"):
        self.num_classes, self.conf_thresh, self.downsample_ratio = num_classes, conf_thresh, downsample_ratio
        self.clip_bbox, self.scale_x_y, self.data_format = clip_bbox, scale_x_y, data_format

    def __call__(self, yolo_head_out_nhwc, anchors, im_shape, scale_factor, var_weight=None):
        origin_shape = (im_shape / scale_factor).to(torch.int32)                      # :560-561
        return E.yolo_box(yolo_head_out_nhwc, anchors, self.num_classes, origin_shape, self.conf_thresh, self.downsample_ratio,
                          self.clip_bbox, self.scale_x_y)


class MultiClassNMS:
    """utils/layers.py:84-129 with the semantics of tlx_multiclass_nms (utils/ops.py:255-329)."""

    def __init__(self, score_threshold=0.05, nms_top_k=-1, keep_top_k=100, nms_threshold=0.5, **kwds):
        self.score_threshold, self.nms_top_k, self.keep_top_k, self.nms_threshold = score_threshold, nms_top_k, keep_top_k, nms_threshold

    def __call__(self, bboxes, score, return_index=False):
        return E.multiclass_nms(bboxes, score, self.score_threshold, self.nms_threshold, self.keep_top_k, return_index=return_index)


def cvt_results(det, cnt):
    """utils/ops.py:397-405 on the (detections, counts) of MultiClassNMS: rows of all images concatenated."""
    cnt_h = cnt.cpu().numpy()
    rows = np.concatenate([det[i, :int(c)].cpu().numpy() for i, c in enumerate(cnt_h)] or [np.zeros((0, 6), np.float32)])
    return dict(labels=rows[:, 0].astype(int), scores=rows[:, 1].astype(float), boxes=rows[:, 2:].astype(int),
                bbox_num=cnt_h.item() if cnt_h.size == 1 else cnt_h)


class YOLOv3(nn.Module):
    def __init__(self, backbone="DarkNet", data_format="channels_first", for_mot=False):
        super().__init__()
        kwds = dict(data_format=data_format)
        self.backbone = create(backbone, **kwds)
        self.neck = YOLOv3FPN(**kwds)
        self.yolo_head = YOLOv3Head(**kwds)
        self.decode = YOLOBox(**kwds)                                                           # :41-42
        self.nms = MultiClassNMS(score_threshold=0.01, nms_threshold=0.5, nms_top_k=1000)       # :43-47
        self.for_mot, self.data_format = for_mot, data_format

    def forward(self, inputs):
        self._require_eval()
        v = as_nhwc(inputs["images"], self.data_format)
        body = self.backbone.run_nhwc(v)
        emb = [] if self.for_mot else None
        neck = self.neck.run_nhwc(body, emb)
        head = self.yolo_head.run_nhwc(neck)
        conv = lambda ts: [from_nhwc(t, self.data_format) for t in ts]
        out = {"images": inputs["images"], "body_feats": conv(body), "neck_feats": conv(neck), "yolo_head_outs": conv(head)}
        if self.for_mot:
            # :64-65: the embedding maps of a tracker (JDE).  The for_mot branch (:70-78) unpacks FOUR values from a post-process
            # the reference does not ship (its BBoxPostProcess returns two, utils/post_process.py:53); what a tracker needs from
            # it — which candidate box each detection row came from, to pick that box's embedding — is "nms_keep_idx" below
            # (tlxmi_multiclass_nms_index).
            out["emb_feats"] = conv(emb)
        # :67-103: decode + NMS -> labels / scores / boxes / bbox_num
        img = inputs["images"]
        n, (h, w) = img.shape[0], (img.shape[2:4] if self.data_format == "channels_first" else img.shape[1:3])
        im_shape = inputs.get("im_shape", torch.tensor([[h, w]] * n, dtype=torch.float32))
        scale_factor = inputs.get("scale_factor", torch.ones_like(torch.as_tensor(im_shape, dtype=torch.float32)))
        bboxes, scores = self.decode(head, self.yolo_head.mask_anchors, torch.as_tensor(im_shape, dtype=torch.float32).cpu(),
                                     torch.as_tensor(scale_factor, dtype=torch.float32).cpu())
        if self.for_mot:
            det, cnt, keep = self.nms(bboxes, scores, return_index=True)
            out["nms_keep_idx"] = keep                          # (N, keep_top_k) int32: row of `bboxes[n]`, -1 past bbox_num[n]
        else:
            det, cnt = self.nms(bboxes, scores)
        out.update(cvt_results(det, cnt))
        out["detections"], out["detection_counts"] = det, cnt
        return out
