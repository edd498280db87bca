"""YOLOv3 = DarkNet-53 -> FPN neck -> 1x1 head convs, forward only, on the MI355X engine.

Mirrors tlxcv/models/detection/yolov3.py: YoloDetBlock :122-183, YOLOv3FPN :186-258, YOLOv3Head :261-378
(head conv :311-322, 3*(num_classes+5) channels, bias).  In the neck, `interpolate(route, scale_factor=2)` +
`tlx.concat([route, x])` (:246-256) are two copy kernels into one pre-sized NHWC buffer (nearest x2 written
at channel offset 0, the backbone map at its channel offset) instead of materialising both.
Out of scope: YOLOv3Loss, Gt2YoloTarget, box decode and NMS — `yolo_box` exists only on
the Paddle backend in the reference (utils/ops.py:436-452).  forward() returns the raw head maps."""
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

    def run_nhwc(self, feats):
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
            if i < self.num_blocks - 1:
                route = self.routes[i].run_nhwc(route)
        return out

    def forward(self, X, for_mot=False):
        if for_mot:
            raise NotImplementedError("for_mot (embedding outputs) is out of scope")
        return [from_nhwc(t, self.data_format) for t in self.run_nhwc([as_nhwc(x, self.data_format) for x in X])]


class YOLOv3Head(nn.Module):
    def __init__(self, in_channels=[1024, 512, 256],
                 anchors=[[10, 13], [16, 30], [33, 23], [30, 61], [62, 45], [59, 119], [116, 90], [156, 198], [373, 326]],
                 anchor_masks=[[6, 7, 8], [3, 4, 5], [0, 1, 2]], num_classes=92, loss=None, batch_transforms=None,
                 iou_aware=False, iou_aware_factor=0.4, data_format="channels_first"):
        super().__init__()
        assert len(in_channels) > 0, "in_channels length should > 0"
        if iou_aware:
            raise NotImplementedError("iou_aware head is out of scope")
        self.in_channels, self.num_classes, self.data_format = in_channels, num_classes, data_format
        self.parse_anchor(anchors, anchor_masks)
        self.num_outputs = len(self.anchors)
        self.yolo_outputs = []
        for i, anc in enumerate(self.anchors):
            self.yolo_outputs.append(nn.GroupConv2d(
                in_channels=self.in_channels[i], out_channels=len(anc) * (self.num_classes + 5), kernel_size=1, stride=1,
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
        return [fn.run_nhwc(f) for fn, f in zip(self.yolo_outputs, feats)]

    def forward(self, outputs, targets=None):
        if targets is not None:
            raise NotImplementedError("YOLOv3Loss is out of scope (training)")
        feats = outputs["neck_feats"] if isinstance(outputs, dict) else outputs
        return [from_nhwc(y, self.data_format) for y in self.run_nhwc([as_nhwc(f, self.data_format) for f in feats])]


class YOLOv3(nn.Module):
    def __init__(self, backbone="DarkNet", data_format="channels_first", for_mot=False):
        super().__init__()
        kwds = dict(data_format=data_format)
        self.backbone = create(backbone, **kwds)
        self.neck = YOLOv3FPN(**kwds)
        self.yolo_head = YOLOv3Head(**kwds)
        self.post_process = None      # decode + NMS: CPU/Paddle-only in the reference, out of scope
        self.for_mot, self.data_format = for_mot, data_format

    def forward(self, inputs):
        self._require_eval()
        v = as_nhwc(inputs["images"], self.data_format)
        body = self.backbone.run_nhwc(v)
        neck = self.neck.run_nhwc(body)
        head = self.yolo_head.run_nhwc(neck)
        conv = lambda ts: [from_nhwc(t, self.data_format) for t in ts]
        return {"images": inputs["images"], "body_feats": conv(body), "neck_feats": conv(neck),
                "yolo_head_outs": conv(head)}
