"""MobileNet (v1) detection backbone on the MI355X engine — same classes / parameter tree as
tlxcv/models/detection/backbones/mobilenet_v1.py:7-240 (the `backbone="MobileNet"` choice of YOLOv3, yolov3.py:6,36).
ConvBNLayer (:41-49: conv without bias, BatchNorm, relu / relu6) is one launch: the 3x3 depthwise convs
(`num_groups == channels`) run tlxmi_dwconv2d, everything else the implicit GEMM, both with the folded BatchNorm +
activation epilogue.  Takes the detectors' dict input ({"images": ...}, :234) and returns the feature maps after
blocks `feature_maps` (default 4, 6, 13 = strides 8 / 16 / 32, :228-240)."""
from numbers import Integral

from .... import engine as E
from ....tlx import nn
from ....tlx.nn import as_nhwc, from_nhwc

__all__ = ["MobileNet"]


class ConvBNLayer(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, num_groups=1, act="relu", conv_lr=1.0,
                 conv_decay=0.0, norm_decay=0.0, norm_type="bn", name=None, data_format="channels_first"):
        super().__init__(name=name)
        self.act = act
        self._conv = nn.GroupConv2d(kernel_size=kernel_size, stride=stride, padding=padding, in_channels=in_channels,
                                    out_channels=out_channels, W_init=nn.initializers.xavier_uniform(), b_init=False,
                                    n_group=num_groups, data_format=data_format)
        if norm_type in ["sync_bn", "bn"]:
            self.my_batch_norm = nn.BatchNorm2d(num_features=out_channels, data_format=data_format)
        else:
            raise NotImplementedError(f"norm_type {norm_type!r}")       # the reference would fail in forward (:42)
        self.data_format = data_format

    def run_nhwc(self, v):
        act = {"relu": E.ACT_RELU, "relu6": E.ACT_RELU6}.get(self.act, E.ACT_NONE)     # :43-46: any other value = no activation
        return self._conv.run_nhwc(v, self.my_batch_norm, act)

    def forward(self, x):
        return from_nhwc(self.run_nhwc(as_nhwc(x, self.data_format)), self.data_format)


class DepthwiseSeparable(nn.Module):
    def __init__(self, in_channels, out_channels1, out_channels2, num_groups, stride, scale, conv_lr=1.0, conv_decay=0.0,
                 norm_decay=0.0, norm_type="bn", name=None, data_format="channels_first"):
        super().__init__(name=name)
        self._depthwise_conv = ConvBNLayer(in_channels, int(out_channels1 * scale), kernel_size=3, stride=stride, padding=1,
                                           num_groups=int(num_groups * scale), norm_type=norm_type, data_format=data_format)
        self._pointwise_conv = ConvBNLayer(int(out_channels1 * scale), int(out_channels2 * scale), kernel_size=1, stride=1,
                                           padding=0, norm_type=norm_type, data_format=data_format)
        self.data_format = data_format

    def run_nhwc(self, v):
        return self._pointwise_conv.run_nhwc(self._depthwise_conv.run_nhwc(v))

    def forward(self, x):
        return from_nhwc(self.run_nhwc(as_nhwc(x, self.data_format)), self.data_format)


class ExtraBlock(nn.Module):
    def __init__(self, in_channels, out_channels1, out_channels2, num_groups=1, stride=2, conv_lr=1.0, conv_decay=0.0,
                 norm_decay=0.0, norm_type="bn", data_format="channels_first", name=None):
        super().__init__(name=name)
        kw = dict(num_groups=int(num_groups), act="relu6", norm_type=norm_type, data_format=data_format)
        self.pointwise_conv = ConvBNLayer(in_channels, int(out_channels1), kernel_size=1, stride=1, padding=0, **kw)
        self.normal_conv = ConvBNLayer(int(out_channels1), int(out_channels2), kernel_size=3, stride=stride, padding=1, **kw)
        self.data_format = data_format

    def run_nhwc(self, v):
        return self.normal_conv.run_nhwc(self.pointwise_conv.run_nhwc(v))

    def forward(self, x):
        return from_nhwc(self.run_nhwc(as_nhwc(x, self.data_format)), self.data_format)


class MobileNet(nn.Module):
    def __init__(self, norm_type="bn", norm_decay=0.0, conv_decay=0.0, scale=1, conv_learning_rate=1.0,
                 feature_maps=[4, 6, 13], with_extra_blocks=False,
                 extra_block_filters=[[256, 512], [128, 256], [128, 256], [64, 128]], data_format="channels_first"):
        super().__init__()
        if isinstance(feature_maps, Integral):
            feature_maps = [feature_maps]
        self.feature_maps, self.with_extra_blocks, self.extra_block_filters = feature_maps, with_extra_blocks, extra_block_filters
        self._out_channels = []
        self.data_format = data_format
        kw = dict(norm_type=norm_type, data_format=data_format)
        self.conv1 = ConvBNLayer(in_channels=3, out_channels=int(32 * scale), kernel_size=3, stride=2, padding=1, **kw)
        # plain Python lists, as in the reference (:204, :220); Module adopts them as dwsl_<i> / extra_blocks_<i>
        self.dwsl = []
        self.cfgs = [[32, 64, 1], [64, 128, 2], [128, 128, 1], [128, 256, 2], [256, 256, 1], [256, 512, 2],
                     *[[512, 512, 1] for _ in range(5)], [512, 1024, 2], [1024, 1024, 1]]          # :205-215
        for _i, _o, _s in self.cfgs:
            self.dwsl.append(DepthwiseSeparable(in_channels=int(_i * scale), out_channels1=_i, out_channels2=_o, num_groups=_i,
                                                stride=_s, scale=scale, **kw))
            self._update_out_channels(int(_o * scale), len(self.dwsl), feature_maps)
        self.extra_blocks = []
        if with_extra_blocks:
            for i, (out0, out1) in enumerate(extra_block_filters):
                in_c = 1024 if i == 0 else extra_block_filters[i - 1][1]
                self.extra_blocks.append(ExtraBlock(in_c, out0, out1, **kw))
                self._update_out_channels(out1, len(self.dwsl) + i + 1, feature_maps)

    def _update_out_channels(self, channel, feature_idx, feature_maps):
        if feature_idx in feature_maps:
            self._out_channels.append(channel)

    def run_nhwc(self, v):
        outs = []
        y = self.conv1.run_nhwc(v)
        for idx, block in enumerate(self.dwsl + self.extra_blocks, start=1):
            y = block.run_nhwc(y)
            if idx in self.feature_maps:
                outs.append(y)
        return outs

    def forward(self, inputs):
        x = inputs["images"]                                             # :234
        return [from_nhwc(o, self.data_format) for o in self.run_nhwc(as_nhwc(x, self.data_format))]
