"""ResNeXt-50/101/152 (32x4d, 64x4d) forward graph on the MI355X engine — same constructor / parameter tree as
tlxcv/models/classification/resnext.py:18-242 (`conv._conv.filters`, `conv.batch_norm.*`,
`bb_<stage>_<i>.conv{0,1,2}._conv.filters`, `.batch_norm.{gamma,beta,moving_mean,moving_var}`, `.short.*`,
`out.weights/biases`).

Fusions: every ConvBNLayer (:18-57, GroupConv2d without bias + BatchNorm(act)) is ONE launch — dense 1x1 / 7x7
through tlxmi_conv2d, the cardinality-32/64 3x3 (:83-91) through tlxmi_group_conv2d (groups merged into
64-channel launch chunks with a block-diagonal filter); the block's `tlx.add(short, conv2)` + `tlx.relu`
(:117-118) ride in conv2's epilogue; the 7x7/2 stem (:150-158) runs on the 2x2 space-to-depth input like
ResNet's; AdaptiveAvgPool2d(1) + reshape + Linear (:188-203) are the global-average-pool kernel and one
GEMM with the bias as epilogue shift."""
import math

import torch

from ... import engine as E
from ...tlx import nn
from ...tlx.nn import as_nhwc
from ...tlx.nn.initializers import xavier_uniform

__all__ = ['ResNeXt', 'resnext50_32x4d', 'resnext50_64x4d', 'resnext101_32x4d', 'resnext101_64x4d',
           'resnext152_32x4d', 'resnext152_64x4d']

_ACT = {None: E.ACT_NONE, 'relu': E.ACT_RELU}


class ConvBNLayer(nn.Module):
    def __init__(self, num_channels, num_filters, filter_size, stride=1, groups=1, act=None, name=None,
                 data_format='channels_first'):
        super().__init__(name)
        self._conv = nn.GroupConv2d(in_channels=num_channels, out_channels=num_filters, kernel_size=filter_size,
                                    stride=stride, padding=(filter_size - 1) // 2, data_format=data_format,
                                    W_init=xavier_uniform(), b_init=(), n_group=groups)
        self.batch_norm = nn.BatchNorm(act=act, num_features=num_filters, moving_mean_init=xavier_uniform(),
                                       moving_var_init=xavier_uniform(), data_format=data_format)
        self.act_code = _ACT[act]
        self.data_format = data_format

    def run_nhwc(self, v, res=None, act=None):
        """conv + BatchNorm (+ the layer's activation, or `act` when the caller fuses a later one) (+ residual)."""
        return self._conv.run_nhwc(v, self.batch_norm, self.act_code if act is None else act, res=res)

    def forward(self, inputs):
        return nn.from_nhwc(self.run_nhwc(as_nhwc(inputs, self.data_format)), self.data_format)


class BottleneckBlock(nn.Module):
    def __init__(self, num_channels, num_filters, stride, cardinality, shortcut=True, name=None,
                 data_format='channels_first'):
        super().__init__(name)
        wide = num_filters * 2 if cardinality == 32 else num_filters
        self.conv0 = ConvBNLayer(num_channels, num_filters, 1, act='relu', name=name + '_branch2a',
                                 data_format=data_format)
        self.conv1 = ConvBNLayer(num_filters, num_filters, 3, groups=cardinality, stride=stride, act='relu',
                                 name=name + '_branch2b', data_format=data_format)
        self.conv2 = ConvBNLayer(num_filters, wide, 1, act=None, name=name + '_branch2c', data_format=data_format)
        if not shortcut:
            self.short = ConvBNLayer(num_channels, wide, 1, stride=stride, name=name + '_branch1',
                                     data_format=data_format)
        self.shortcut = shortcut
        self.data_format = data_format

    def forward_nhwc(self, v):
        short = v if self.shortcut else self.short.run_nhwc(v)
        y = self.conv1.run_nhwc(self.conv0.run_nhwc(v))
        return self.conv2.run_nhwc(y, res=short, act=E.ACT_RELU)      # add + relu in the epilogue, :117-118

    # -- the block cut at the seam the engine fuses (as resnet.py's BottleneckBlock): run_head() = conv0 -> grouped conv1; the
    #    expand conv2 + skip + relu then runs either alone (finish) or in ONE launch with the next block's conv0
    #    (E.bottleneck_seam: the wide map between two blocks is written once and not re-read by the reduce conv)
    def run_head(self, v, t0=None):
        return self.conv1.run_nhwc(t0 if t0 is not None else self.conv0.run_nhwc(v))

    def skip(self, v):
        return v if self.shortcut else self.short.run_nhwc(v)

    def finish(self, y, short):
        return self.conv2.run_nhwc(y, res=short, act=E.ACT_RELU)

    def seam_with(self, nxt, y, v):
        """(block output, nxt.conv0's output) in one launch, or None when the library has no fused kernel for these layers
        (32x4d: 128 -> 256 -> 128 / 256 and 256 -> 512 -> 256, i.e. stages 1 and 2; fp16, >= 12 images)."""
        c3, c1 = self.conv2._conv, nxt.conv0._conv
        dt = E.precision()
        if (not E.option("seams") or dt != torch.float16 or c3.n_group != 1 or c1.n_group != 1 or c1.stride != (1, 1) or c3.biases is not None
                or c1.biases is not None or self.conv2.act_code != E.ACT_NONE or nxt.conv0.act_code != E.ACT_RELU
                or not E.bottleneck_seam_supported(c3.in_channels, c3.out_channels, c1.out_channels, dt)
                or (c3.in_channels >= 256 and not E.option("seam256"))):
            return None
        n_img = y.shape[0]
        if n_img < 12 or (c3.in_channels >= 256 and n_img < 96 and not E.in_halves()):
            return None
        pk3 = c3._cached("pk", lambda: E.PackedFilter(c3.filters, dt))
        pk1 = c1._cached("pk", lambda: E.PackedFilter(c1.filters, dt))
        bn3, bn1 = self.conv2.batch_norm, nxt.conv0.batch_norm
        s3, h3 = c3._cached(("bn", id(bn3)), lambda: bn3.folded(None), deps=(bn3,))
        s1, h1 = c1._cached(("bn", id(bn1)), lambda: bn1.folded(None), deps=(bn1,))
        return E.bottleneck_seam(y, pk3, s3, h3, self.skip(v), pk1, s1, h1)

    def forward(self, inputs):
        return nn.from_nhwc(self.forward_nhwc(as_nhwc(inputs, self.data_format)), self.data_format)


class ResNeXt(nn.Module):
    def __init__(self, layers=50, num_classes=1000, cardinality=32, input_image_channel=3, name=None,
                 data_format='channels_first'):
        super().__init__(name)
        self.layers = layers
        self.cardinality = cardinality
        self.data_format = data_format
        supported_layers = [50, 101, 152]
        assert layers in supported_layers, 'supported layers are {} but input layer is {}'.format(
            supported_layers, layers)
        supported_cardinality = [32, 64]
        assert cardinality in supported_cardinality, 'supported cardinality is {} but input cardinality is {}'.format(
            supported_cardinality, cardinality)
        depth = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}[layers]
        num_channels = [64, 256, 512, 1024]
        num_filters = [128, 256, 512, 1024] if cardinality == 32 else [256, 512, 1024, 2048]
        self.conv = ConvBNLayer(input_image_channel, 64, 7, stride=2, act='relu', name='res_conv1',
                                data_format=data_format)
        self.pool2d_max = nn.MaxPool2d(kernel_size=3, stride=2, padding=1, data_format=data_format)
        self.block_list = []
        for block in range(len(depth)):
            shortcut = False
            for i in range(depth[block]):
                if layers in [101, 152] and block == 2:
                    conv_name = 'res' + str(block + 2) + ('a' if i == 0 else 'b' + str(i))
                else:
                    conv_name = 'res' + str(block + 2) + chr(97 + i)
                bb = BottleneckBlock(
                    num_channels=num_channels[block] if i == 0 else num_filters[block] * int(64 // self.cardinality),
                    num_filters=num_filters[block], stride=2 if i == 0 and block != 0 else 1,
                    cardinality=self.cardinality, shortcut=shortcut, name=conv_name, data_format=data_format)
                setattr(self, 'bb_%d_%d' % (block, i), bb)
                self.block_list.append(bb)
                shortcut = True
        self.pool2d_avg = nn.AdaptiveAvgPool2d(1, data_format=data_format)
        self.pool2d_avg_channels = num_channels[-1] * 2
        self.out = nn.Linear(in_features=self.pool2d_avg_channels, out_features=num_classes, b_init=xavier_uniform())

    @E.two_streams(128, plan="half")
    def forward(self, inputs):
        x = inputs
        if (self.data_format == 'channels_first' and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
                and not x.permute(0, 2, 3, 1).is_contiguous()):
            v = self.conv._conv.run_stem(x, 2, self.conv.batch_norm, E.ACT_RELU)        # :206
        else:
            v = self.conv.run_nhwc(as_nhwc(x, self.data_format))
        v = self.pool2d_max.run_nhwc(v)                                                   # :207
        t0 = None
        for i, block in enumerate(self.block_list):                                       # :208-209, seams between blocks fused
            y = block.run_head(v, t0)
            t0 = None
            nxt = self.block_list[i + 1] if i + 1 < len(self.block_list) else None
            fused = block.seam_with(nxt, y, v) if nxt is not None else None
            if fused is not None:
                v, t0 = fused
            else:
                v = block.finish(y, block.skip(v))
        v = E.global_avgpool(v)                                                           # :210-211
        return self.out.run(v)


def _resnext(arch, layers, cardinality, pretrained, **kwargs):
    if pretrained:
        raise NotImplementedError("pretrained weights are not bundled; use model.load_weights(...)")
    return ResNeXt(layers=layers, cardinality=cardinality, **kwargs)


def resnext50_32x4d(pretrained=False, **kwargs):
    return _resnext('resnext50_32x4d', 50, 32, pretrained, **kwargs)


def resnext50_64x4d(pretrained=False, **kwargs):
    return _resnext('resnext50_64x4d', 50, 64, pretrained, **kwargs)


def resnext101_32x4d(pretrained=False, **kwargs):
    return _resnext('resnext101_32x4d', 101, 32, pretrained, **kwargs)


def resnext101_64x4d(pretrained=False, **kwargs):
    return _resnext('resnext101_64x4d', 101, 64, pretrained, **kwargs)


def resnext152_32x4d(pretrained=False, **kwargs):
    return _resnext('resnext152_32x4d', 152, 32, pretrained, **kwargs)


def resnext152_64x4d(pretrained=False, **kwargs):
    return _resnext('resnext152_64x4d', 152, 64, pretrained, **kwargs)
