"""ResNeSt-50 / 50-fast / 101 forward graph on the MI355X engine — same constructors / parameter tree as
tlxcv/models/classification/resnest.py:12-747 (`stem.conv1._conv.filters`, `layer2.layer2_bottleneck_0.conv2.conv1._conv.filters`,
`...conv2.conv3.filters`, `...conv4.filters`, `...batch_norm.gamma`, `out.weights`).

Fusions: every ConvBNLayer (:12-51) is ONE launch (dense convs through tlxmi_conv2d, the radix-grouped 3x3 of SplatConv
:101-110 through tlxmi_group_conv2d); SplatConv (:147-166) = that conv, `tlxmi_radix_gap` (split + add_n + global average
pool), two tiny 1x1 convs on the pooled vector, and `tlxmi_split_attention` (rSoftmax :53-82 + split + multiply + add_n in
one pass over the map); the anti-aliasing AvgPool2d layers (:212-218, 250-256, 271-286) are `tlxmi_avgpool2d`; the
shortcut's conv4 + BatchNorm (:287-309) is one launch and the block's add + relu (:324-326) ride in conv3's epilogue."""
import math
from collections import OrderedDict

import torch

from ... import engine as E
from ...tlx import nn
from ...tlx.nn import as_nhwc, from_nhwc
from ...tlx.nn.initializers import xavier_uniform

__all__ = ['ResNeSt', 'resnest50_fast_1s1x64d', 'resnest50', 'resnest101']

_ACT = {None: E.ACT_NONE, 'relu': E.ACT_RELU}


class ConvBNLayer(nn.Module):
    def __init__(self, num_channels, num_filters, filter_size, stride=1, dilation=1, groups=1, act=None,
                 data_format='channels_first', name=None):
        super().__init__(name)
        self._conv = nn.GroupConv2d(in_channels=num_channels, out_channels=num_filters, kernel_size=filter_size,
                                    stride=stride, padding=(filter_size - 1) // 2, dilation=dilation,
                                    W_init=xavier_uniform(), b_init=(), n_group=groups, data_format=data_format)
        self.batch_norm = nn.BatchNorm(act=act, num_features=num_filters, moving_mean_init=xavier_uniform(),
                                       moving_var_init=xavier_uniform(), data_format=data_format)
        self.act_code = _ACT[act]
        self.data_format = data_format

    def run_nhwc(self, v, res=None, act=None):
        return self._conv.run_nhwc(v, self.batch_norm, self.act_code if act is None else act, res=res)

    def forward(self, x):
        return from_nhwc(self.run_nhwc(as_nhwc(x, self.data_format)), self.data_format)


class rSoftmax(nn.Module):
    """resnest.py:53-82; SplatConv.run_nhwc applies it inside tlxmi_split_attention."""

    def __init__(self, radix, cardinality, data_format='channels_first'):
        super().__init__()
        self.radix, self.cardinality, self.data_format = radix, cardinality, data_format


class SplatConv(nn.Module):
    def __init__(self, in_channels, channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True, radix=2,
                 reduction_factor=4, rectify_avg=False, data_format='channels_first', name=None):
        super().__init__(name)
        self.radix, self.cardinality, self.channels = radix, groups, channels
        self.conv1 = ConvBNLayer(in_channels, channels * radix, kernel_size, stride=stride, groups=groups * radix, act='relu',
                                 data_format=data_format, name=name + '_1_weights')
        self.avg_pool2d = nn.AdaptiveAvgPool2d(1, data_format=data_format)
        inter_channels = int(max(in_channels * radix // reduction_factor, 32))
        self.conv2 = ConvBNLayer(channels, inter_channels, 1, stride=1, groups=groups, act='relu', data_format=data_format,
                                 name=name + '_2_weights')
        self.conv3 = nn.GroupConv2d(in_channels=inter_channels, out_channels=channels * radix, kernel_size=1, stride=1,
                                    padding=0, W_init=xavier_uniform(), b_init=(), n_group=groups, data_format=data_format)
        self.rsoftmax = rSoftmax(radix=radix, cardinality=groups, data_format=data_format)
        self.data_format = data_format

    def run_nhwc(self, v):
        x1 = self.conv1.run_nhwc(v)                                            # :148
        N = x1.shape[0]
        gap = E.radix_gap(x1, self.radix).view(N, 1, 1, self.channels)         # :149-155
        att = self.conv3.run_nhwc(self.conv2.run_nhwc(gap))                    # :156-157
        return E.split_attention(x1, att.view(N, self.channels * self.radix), self.radix, self.cardinality)   # :158-165

    def forward(self, x):
        return from_nhwc(self.run_nhwc(as_nhwc(x, self.data_format)), self.data_format)


class BottleneckBlock(nn.Module):
    def __init__(self, inplanes, planes, stride=1, radix=1, cardinality=1, bottleneck_width=64, avd=False, avd_first=False,
                 dilation=1, is_first=False, rectify_avg=False, last_gamma=False, avg_down=False,
                 data_format='channels_first', name=None):
        super().__init__(name)
        self.inplanes, self.planes, self.stride, self.radix, self.cardinality = inplanes, planes, stride, radix, cardinality
        self.avd, self.avd_first, self.dilation, self.is_first, self.avg_down = avd, avd_first, dilation, is_first, avg_down
        self.data_format = data_format
        group_width = int(planes * (bottleneck_width / 64.0)) * cardinality
        self.conv1 = ConvBNLayer(self.inplanes, group_width, 1, stride=1, groups=1, act='relu', data_format=data_format,
                                 name=name + '_conv1')
        if avd and avd_first and (stride > 1 or is_first):
            self.avg_pool2d_1 = nn.AvgPool2d(kernel_size=3, stride=stride, padding=1, data_format=data_format)
        if radix >= 1:
            self.conv2 = SplatConv(group_width, group_width, 3, stride=1, padding=dilation, dilation=dilation,
                                   groups=cardinality, bias=False, radix=radix, rectify_avg=rectify_avg,
                                   data_format=data_format, name=name + '_splat')
        else:
            self.conv2 = ConvBNLayer(group_width, group_width, 3, stride=1, dilation=dilation, groups=cardinality, act='relu',
                                     data_format=data_format, name=name + '_conv2')
        if avd and avd_first == False and (stride > 1 or is_first):  # noqa: E712
            self.avg_pool2d_2 = nn.AvgPool2d(kernel_size=3, stride=stride, padding=1, data_format=data_format)
        self.conv3 = ConvBNLayer(group_width, planes * 4, 1, stride=1, groups=1, act=None, data_format=data_format,
                                 name=name + '_conv3')
        if stride != 1 or self.inplanes != self.planes * 4:
            if avg_down:
                if dilation == 1:
                    self.avg_pool2d_3 = nn.AvgPool2d(kernel_size=stride, stride=stride, padding=0, data_format=data_format)
                else:
                    self.avg_pool2d_3 = nn.AvgPool2d(kernel_size=1, stride=1, padding=0, ceil_mode=True, data_format=data_format)
                self.conv4 = nn.GroupConv2d(in_channels=self.inplanes, out_channels=planes * 4, kernel_size=1, stride=1,
                                            padding=0, W_init=xavier_uniform(), b_init=(), n_group=1, data_format=data_format)
            else:
                self.conv4 = nn.GroupConv2d(in_channels=self.inplanes, out_channels=planes * 4, kernel_size=1, stride=stride,
                                            padding=0, W_init=xavier_uniform(), b_init=(), n_group=1, data_format=data_format)
            self.batch_norm = nn.BatchNorm(act=None, num_features=planes * 4, moving_mean_init=xavier_uniform(),
                                           moving_var_init=xavier_uniform(), data_format=data_format)

    def forward_nhwc(self, v):
        short = v
        pool = self.stride > 1 or self.is_first
        y = self.conv1.run_nhwc(v)                                             # :312
        if self.avd and self.avd_first and pool:
            y = self.avg_pool2d_1.run_nhwc(y)                                  # :313-314
        y = self.conv2.run_nhwc(y)                                             # :315
        if self.avd and self.avd_first == False and pool:  # noqa: E712
            y = self.avg_pool2d_2.run_nhwc(y)                                  # :316-317
        if self.stride != 1 or self.inplanes != self.planes * 4:               # :319-323
            if self.avg_down:
                short = self.avg_pool2d_3.run_nhwc(short)
            short = self.conv4.run_nhwc(short, self.batch_norm)
        return self.conv3.run_nhwc(y, res=short, act=E.ACT_RELU)               # :318, :324-326

    # -- the block cut at the seam the engine fuses (as resnet.py's BottleneckBlock): run_head() = conv1 -> (pool) -> split-attention
    #    conv -> (pool); the expand conv3 + skip + relu then runs alone (finish) or in ONE launch with the next block's conv1
    def run_head(self, v, t1=None):
        pool = self.stride > 1 or self.is_first
        y = t1 if t1 is not None else self.conv1.run_nhwc(v)
        if self.avd and self.avd_first and pool:
            y = self.avg_pool2d_1.run_nhwc(y)
        y = self.conv2.run_nhwc(y)
        if self.avd and self.avd_first == False and pool:  # noqa: E712
            y = self.avg_pool2d_2.run_nhwc(y)
        return y

    def skip(self, v):
        if self.stride != 1 or self.inplanes != self.planes * 4:
            if self.avg_down:
                v = self.avg_pool2d_3.run_nhwc(v)
            return self.conv4.run_nhwc(v, self.batch_norm)
        return v

    def finish(self, y, short):
        return self.conv3.run_nhwc(y, res=short, act=E.ACT_RELU)

    def seam_with(self, nxt, y, v):
        """(block output, nxt.conv1's output) in one launch, or None (no fused kernel for these widths / fp32 / few images)."""
        c3, c1 = self.conv3._conv, nxt.conv1._conv
        dt = E.precision()
        if (not E.option("seams") or dt != torch.float16 or c3.n_group != 1 or c1.n_group != 1 or c1.stride != (1, 1) or c3.biases is not None
                or c1.biases is not None or self.conv3.act_code != E.ACT_NONE or nxt.conv1.act_code != E.ACT_RELU
                or not E.bottleneck_seam_supported(c3.in_channels, c3.out_channels, c1.out_channels, dt)
                or (c3.in_channels >= 256 and not E.option("seam256"))):
            return None
        n_img = y.shape[0]
        if n_img < 12 or (c3.in_channels >= 256 and n_img < 96 and not E.in_halves()):
            return None
        pk3 = c3._cached("pk", lambda: E.PackedFilter(c3.filters, dt))
        pk1 = c1._cached("pk", lambda: E.PackedFilter(c1.filters, dt))
        bn3, bn1 = self.conv3.batch_norm, nxt.conv1.batch_norm
        s3, h3 = c3._cached(("bn", id(bn3)), lambda: bn3.folded(None), deps=(bn3,))
        s1, h1 = c1._cached(("bn", id(bn1)), lambda: bn1.folded(None), deps=(bn1,))
        return E.bottleneck_seam(y, pk3, s3, h3, self.skip(v), pk1, s1, h1)

    def forward(self, x):
        return from_nhwc(self.forward_nhwc(as_nhwc(x, self.data_format)), self.data_format)


def run_block_chain(blocks, v):
    """All bottleneck blocks of the four stages in order, every block-to-block seam fused where the library has the kernel."""
    t1 = None
    for i, blk in enumerate(blocks):
        y = blk.run_head(v, t1)
        t1 = None
        nxt = blocks[i + 1] if i + 1 < len(blocks) else None
        fused = blk.seam_with(nxt, y, v) if nxt is not None else None
        if fused is not None:
            v, t1 = fused
        else:
            v = blk.finish(y, blk.skip(v))
    return v


class ResNeStLayer(nn.Module):
    def __init__(self, inplanes, planes, blocks, radix, cardinality, bottleneck_width, avg_down, avd, avd_first, rectify_avg,
                 last_gamma, stride=1, dilation=1, is_first=True, data_format='channels_first', name=None):
        super().__init__(name)
        self.inplanes, self.planes, self.blocks = inplanes, planes, blocks
        if dilation not in (1, 2, 4):
            raise RuntimeError('=>unknown dilation size')
        common = dict(radix=radix, cardinality=cardinality, bottleneck_width=bottleneck_width, avg_down=avg_down, avd=avd,
                      avd_first=avd_first, rectify_avg=rectify_avg, last_gamma=last_gamma, data_format=data_format)
        first = name + '_bottleneck_0'
        setattr(self, first, BottleneckBlock(inplanes=self.inplanes, planes=planes, stride=stride,
                                             dilation=1 if dilation in (1, 2) else 2, is_first=is_first, name=first, **common))
        self.inplanes = planes * 4
        self.bottleneck_block_list = [getattr(self, first)]
        for i in range(1, blocks):
            curr_name = name + '_bottleneck_' + str(i)
            setattr(self, curr_name, BottleneckBlock(inplanes=self.inplanes, planes=planes, dilation=dilation, name=curr_name,
                                                     **common))
            self.bottleneck_block_list.append(getattr(self, curr_name))

    def forward_nhwc(self, v):
        for blk in self.bottleneck_block_list:
            v = blk.forward_nhwc(v)
        return v

    def forward(self, x):
        for blk in self.bottleneck_block_list:
            x = blk(x)
        return x


class ResNeSt(nn.Module):
    def __init__(self, layers, radix=1, groups=1, bottleneck_width=64, dilated=False, dilation=1, deep_stem=False,
                 stem_width=64, avg_down=False, rectify_avg=False, avd=False, avd_first=False, final_drop=0.0,
                 last_gamma=False, num_classes=1000, data_format='channels_first', name=None):
        super().__init__(name)
        self.cardinality, self.bottleneck_width = groups, bottleneck_width
        self.inplanes = stem_width * 2 if deep_stem else 64
        self.radix, self.avd, self.avd_first, self.deep_stem, self.stem_width = radix, avd, avd_first, deep_stem, stem_width
        self.layers, self.final_drop, self.dilated, self.dilation = layers, final_drop, dilated, dilation
        self.data_format = data_format
        if dilated or dilation != 1:
            raise NotImplementedError("dilated ResNeSt variants are not exported by the reference's constructors (resnest.py:692-735)")
        if deep_stem:
            self.stem = nn.Sequential(OrderedDict([
                ('conv1', ConvBNLayer(3, stem_width, 3, stride=2, act='relu', data_format=data_format, name='conv1')),
                ('conv2', ConvBNLayer(stem_width, stem_width, 3, stride=1, act='relu', data_format=data_format, name='conv2')),
                ('conv3', ConvBNLayer(stem_width, stem_width * 2, 3, stride=1, act='relu', data_format=data_format, name='conv3')),
            ]))
        else:
            self.stem = ConvBNLayer(3, stem_width, 7, stride=2, act='relu', data_format=data_format, name='conv1')
        self.max_pool2d = nn.MaxPool2d(kernel_size=3, stride=2, padding=1, data_format=data_format)
        common = dict(radix=radix, cardinality=groups, bottleneck_width=bottleneck_width, avg_down=avg_down, avd=avd,
                      avd_first=avd_first, rectify_avg=rectify_avg, last_gamma=last_gamma, data_format=data_format)
        self.layer1 = ResNeStLayer(inplanes=self.inplanes, planes=64, blocks=layers[0], stride=1, dilation=1, is_first=False,
                                   name='layer1', **common)
        self.layer2 = ResNeStLayer(inplanes=256, planes=128, blocks=layers[1], stride=2, name='layer2', **common)
        self.layer3 = ResNeStLayer(inplanes=512, planes=256, blocks=layers[2], stride=2, name='layer3', **common)
        self.layer4 = ResNeStLayer(inplanes=1024, planes=512, blocks=layers[3], stride=2, name='layer4', **common)
        self.pool2d_avg = nn.AdaptiveAvgPool2d(1, data_format=data_format)
        self.out_channels = 2048
        self.out = nn.Linear(in_features=self.out_channels, out_features=num_classes, b_init=xavier_uniform())

    @E.two_streams(128, plan="full")
    def forward(self, x):
        first = list(self.stem)[0] if self.deep_stem else self.stem
        rest = list(self.stem)[1:] if self.deep_stem else []
        if (self.data_format == 'channels_first' and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
                and not x.permute(0, 2, 3, 1).is_contiguous()):
            v = first._conv.run_stem(x, 2, first.batch_norm, E.ACT_RELU)        # 3x3/2 (7x7/2) on the 2x2 space-to-depth image
        else:
            nchw = x if self.data_format == 'channels_first' else x.permute(0, 3, 1, 2)
            v = first.run_nhwc(E.nchw_to_nhwc(nchw, E.precision()))
        for m in rest:
            v = m.run_nhwc(v)
        v = self.max_pool2d.run_nhwc(v)                                          # :681
        blocks = [b for layer in (self.layer1, self.layer2, self.layer3, self.layer4) for b in layer.bottleneck_block_list]
        v = run_block_chain(blocks, v)                                           # :682-685, seams between blocks fused (fp16)
        v = E.global_avgpool(v)                                                  # :686-687
        return self.out.run(v)                                                   # :688


def _resnest(arch, pretrained, **kwargs):
    if pretrained:
        raise NotImplementedError("pretrained weights are not bundled; use model.load_weights(...)")
    cfg = {'resnest50_fast_1s1x64d': dict(layers=[3, 4, 6, 3], radix=1, stem_width=32, avd_first=True),
           'resnest50': dict(layers=[3, 4, 6, 3], radix=2, stem_width=32, avd_first=False),
           'resnest101': dict(layers=[3, 4, 23, 3], radix=2, stem_width=64, avd_first=False)}[arch]
    return ResNeSt(groups=1, deep_stem=True, avg_down=True, avd=True, final_drop=0.0, **cfg, **kwargs)


def resnest50_fast_1s1x64d(pretrained=False, **kwargs):
    return _resnest('resnest50_fast_1s1x64d', pretrained, **kwargs)


def resnest50(pretrained=False, **kwargs):
    return _resnest('resnest50', pretrained, **kwargs)


def resnest101(pretrained=False, **kwargs):
    return _resnest('resnest101', pretrained, **kwargs)
