"""MobileNetV2 / MobileNetV3 forward graphs on the MI355X engine.

Mirror tlxcv/models/classification/mobilenetv2.py:15-148, mobilenetv3.py:21-351, ops/ops_fusion.py:11-48 and
utils/common_func.py:1-16 (Paddle-converted files in the reference: restated from their text).
Per inverted-residual block: 1x1 expand (+BN+act) = implicit GEMM; k3/k5 depthwise (+BN+act) = HBM-bound
dwconv kernel; Squeeze-Excitation = global-avgpool -> two tiny GEMMs (ReLU / HardSigmoid epilogues) ->
one gating pass (tlxmi_scale_channels); 1x1 linear projection (+BN) = implicit GEMM with the block's
residual add fused into its epilogue (mobilenetv2.py:36-40, mobilenetv3.py:111-121)."""
from functools import partial

from ... import engine as E
from ... import tlx
from ...tlx import nn
from ...tlx.nn import as_nhwc, from_nhwc
from .mobilenetv1 import ConvNormActivation

__all__ = ["MobileNetV2", "mobilenet_v2", "MobileNetV3Small", "MobileNetV3Large", "mobilenet_v3_small",
           "mobilenet_v3_large"]


def _make_divisible(v, divisor=8, min_value=None):
    """utils/common_func.py:1-16."""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


class _NHWCBlock(nn.Module):
    def forward(self, x):
        return from_nhwc(self.run_nhwc(as_nhwc(x, 'channels_first')), 'channels_first')


class InvertedResidual(_NHWCBlock):
    """mobilenetv2.py:15-40."""

    def __init__(self, inp, oup, stride, expand_ratio, batch_norm=nn.BatchNorm2d):
        super().__init__()
        self.stride = stride
        assert stride in [1, 2]
        hidden_dim = int(round(inp * expand_ratio))
        self.use_res_connect = self.stride == 1 and inp == oup
        layers = []
        if expand_ratio != 1:
            layers.append(ConvNormActivation(inp, hidden_dim, kernel_size=1, batch_norm=batch_norm,
                                             activation_layer=nn.ReLU6))
        layers.extend([
            ConvNormActivation(hidden_dim, hidden_dim, stride=stride, groups=hidden_dim, batch_norm=batch_norm,
                               activation_layer=nn.ReLU6),
            nn.GroupConv2d(in_channels=hidden_dim, out_channels=oup, kernel_size=1, stride=1, padding=0, b_init=(),
                           W_init=nn.initializers.HeNormal(), data_format='channels_first'),
            batch_norm(num_features=oup, data_format='channels_first')])
        self.conv = nn.Sequential([*layers])

    def run_nhwc(self, v):
        mods = list(self.conv)
        h = v
        for m in mods[:-2]:
            h = m.run_nhwc(h)
        return mods[-2].run_nhwc(h, mods[-1], res=v if self.use_res_connect else None)   # x + conv(x), :37-38


class MobileNetV2(nn.Module):
    """mobilenetv2.py:43-109."""

    def __init__(self, scale=1.0, num_classes=1000, with_pool=True):
        super().__init__()
        self.num_classes, self.with_pool = num_classes, with_pool
        input_channel, last_channel, round_nearest = 32, 1280, 8
        setting = [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1], [6, 160, 3, 2],
                   [6, 320, 1, 1]]
        input_channel = _make_divisible(input_channel * scale, round_nearest)
        self.last_channel = _make_divisible(last_channel * max(1.0, scale), round_nearest)
        features = [ConvNormActivation(3, input_channel, stride=2, activation_layer=nn.ReLU6)]
        for t, c, n, s in setting:
            output_channel = _make_divisible(c * scale, round_nearest)
            for i in range(n):
                features.append(InvertedResidual(input_channel, output_channel, s if i == 0 else 1, expand_ratio=t))
                input_channel = output_channel
        features.append(ConvNormActivation(input_channel, self.last_channel, kernel_size=1, activation_layer=nn.ReLU6))
        self.features = nn.Sequential([*features])
        if with_pool:
            self.pool2d_avg = nn.AdaptiveAvgPool2d(1, data_format='channels_first')
        if self.num_classes > 0:
            self.classifier = nn.Sequential([nn.Dropout(0.2), nn.Linear(in_features=self.last_channel,
                                                                        out_features=num_classes)])

    @E.two_streams(128, plan="full")
    def forward(self, x):
        v = as_nhwc(x, 'channels_first')
        for f in self.features:
            v = f.run_nhwc(v)
        if self.with_pool:
            v = E.global_avgpool(v)
            if self.num_classes > 0:
                return self.classifier[1].run(v)
            return v.view(v.shape[0], -1, 1, 1)
        y = from_nhwc(v, 'channels_first')
        return self.classifier(tlx.flatten(y.contiguous(), 1)) if self.num_classes > 0 else y


def mobilenet_v2(pretrained=False, scale=1.0, **kwargs):
    if pretrained:
        raise NotImplementedError("pretrained weights are not bundled; use model.load_weights(...)")
    return MobileNetV2(scale=scale, **kwargs)


# ---------------------------------------------------------------------------------------------
class SqueezeExcitation(_NHWCBlock):
    """mobilenetv3.py:21-56: scale = sigma(fc2(delta(fc1(avgpool(x))))); return scale * x."""

    def __init__(self, input_channels, squeeze_channels, activation=nn.ReLU, scale_activation=nn.Sigmoid):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d(1, data_format='channels_first')
        self.fc1 = nn.GroupConv2d(in_channels=input_channels, out_channels=squeeze_channels, kernel_size=1, padding=0,
                                  W_init=nn.initializers.HeNormal(), data_format='channels_first')
        self.fc2 = nn.GroupConv2d(in_channels=squeeze_channels, out_channels=input_channels, kernel_size=1, padding=0,
                                  W_init=nn.initializers.HeNormal(), data_format='channels_first')
        self.activation = activation()
        self.scale_activation = scale_activation()

    def run_nhwc(self, v):
        N, Cc = v.shape[0], v.shape[-1]
        s = E.global_avgpool(v).view(N, 1, 1, Cc)
        s = self.fc1.run_nhwc(s, act=self.activation.ACT)
        s = self.fc2.run_nhwc(s, act=self.scale_activation.ACT)
        return E.scale_channels(v, s.view(N, Cc))


class InvertedResidualConfig:
    def __init__(self, in_channels, kernel, expanded_channels, out_channels, use_se, activation, stride, scale=1.0):
        self.in_channels = self.adjust_channels(in_channels, scale=scale)
        self.kernel = kernel
        self.expanded_channels = self.adjust_channels(expanded_channels, scale=scale)
        self.out_channels = self.adjust_channels(out_channels, scale=scale)
        self.use_se = use_se
        if activation is None:
            self.activation_layer = None
        elif activation == 'relu':
            self.activation_layer = nn.ReLU
        elif activation == 'hardswish':
            self.activation_layer = nn.Hardswish
        else:
            raise RuntimeError('The activation function is not supported: {}'.format(activation))
        self.stride = stride

    @staticmethod
    def adjust_channels(channels, scale=1.0):
        return _make_divisible(channels * scale, 8)


class InvertedResidualV3(_NHWCBlock):
    """mobilenetv3.py:84-121."""

    def __init__(self, in_channels, expanded_channels, out_channels, filter_size, stride, use_se, activation_layer,
                 batch_norm):
        super().__init__()
        self.use_res_connect = stride == 1 and in_channels == out_channels
        self.use_se = use_se
        self.expand = in_channels != expanded_channels
        if self.expand:
            self.expand_conv = ConvNormActivation(in_channels=in_channels, out_channels=expanded_channels,
                                                  kernel_size=1, stride=1, padding=0, batch_norm=batch_norm,
                                                  activation_layer=activation_layer)
        self.bottleneck_conv = ConvNormActivation(in_channels=expanded_channels, out_channels=expanded_channels,
                                                  kernel_size=filter_size, stride=stride,
                                                  padding=int((filter_size - 1) // 2), groups=expanded_channels,
                                                  batch_norm=batch_norm, activation_layer=activation_layer)
        if self.use_se:
            self.mid_se = SqueezeExcitation(expanded_channels, _make_divisible(expanded_channels // 4),
                                            scale_activation=nn.HardSigmoid)
        self.linear_conv = ConvNormActivation(in_channels=expanded_channels, out_channels=out_channels, kernel_size=1,
                                              stride=1, padding=0, batch_norm=batch_norm, activation_layer=None)

    def run_nhwc(self, v):
        h = self.expand_conv.run_nhwc(v) if self.expand else v
        h = self.bottleneck_conv.run_nhwc(h)
        if self.use_se:
            h = self.mid_se.run_nhwc(h)
        return self.linear_conv.run_nhwc(h, res=v if self.use_res_connect else None)      # :119-120


class MobileNetV3(nn.Module):
    """mobilenetv3.py:124-180 (BatchNorm epsilon 1e-3, momentum 0.99, :148)."""

    def __init__(self, config, last_channel, scale=1.0, num_classes=1000, with_pool=True):
        super().__init__()
        self.config, self.scale, self.last_channel = config, scale, last_channel
        self.num_classes, self.with_pool = num_classes, with_pool
        self.firstconv_in_channels = config[0].in_channels
        self.lastconv_in_channels = config[-1].in_channels
        self.lastconv_out_channels = self.lastconv_in_channels * 6
        batch_norm = partial(nn.BatchNorm2d, epsilon=0.001, momentum=0.99)
        self.conv = ConvNormActivation(in_channels=3, out_channels=self.firstconv_in_channels, kernel_size=3, stride=2,
                                       padding=1, groups=1, activation_layer=nn.Hardswish, batch_norm=batch_norm)
        self.blocks = nn.Sequential([*[
            InvertedResidualV3(in_channels=c.in_channels, expanded_channels=c.expanded_channels,
                               out_channels=c.out_channels, filter_size=c.kernel, stride=c.stride, use_se=c.use_se,
                               activation_layer=c.activation_layer, batch_norm=batch_norm) for c in self.config]])
        self.lastconv = ConvNormActivation(in_channels=self.lastconv_in_channels, out_channels=self.lastconv_out_channels,
                                           kernel_size=1, stride=1, padding=0, groups=1, batch_norm=batch_norm,
                                           activation_layer=nn.Hardswish)
        if with_pool:
            self.avgpool = nn.AdaptiveAvgPool2d(1, data_format='channels_first')
        if num_classes > 0:
            self.classifier = nn.Sequential([
                nn.Linear(in_features=self.lastconv_out_channels, out_features=self.last_channel), nn.Hardswish(),
                nn.Dropout(p=0.2), nn.Linear(in_features=self.last_channel, out_features=num_classes)])

    @E.two_streams(128, plan="full", eager=False)      # MobileNetV3: ~200 tiny launches, host-bound kernel by kernel
    def forward(self, x):
        v = self.conv.run_nhwc(as_nhwc(x, 'channels_first'))
        for b in self.blocks:
            v = b.run_nhwc(v)
        v = self.lastconv.run_nhwc(v)
        if self.with_pool:
            v = E.global_avgpool(v)
            if self.num_classes > 0:
                h = self.classifier[0].run(v, act=E.ACT_HARDSWISH)            # Linear + Hardswish, :166-168
                return self.classifier[3].run(h)
            return v.view(v.shape[0], -1, 1, 1)
        y = from_nhwc(v, 'channels_first')
        return self.classifier(tlx.flatten(y.contiguous(), 1)) if self.num_classes > 0 else y


class MobileNetV3Small(MobileNetV3):
    def __init__(self, scale=1.0, num_classes=1000, with_pool=True):
        C = InvertedResidualConfig
        config = [C(16, 3, 16, 16, True, 'relu', 2, scale), C(16, 3, 72, 24, False, 'relu', 2, scale),
                  C(24, 3, 88, 24, False, 'relu', 1, scale), C(24, 5, 96, 40, True, 'hardswish', 2, scale),
                  C(40, 5, 240, 40, True, 'hardswish', 1, scale), C(40, 5, 240, 40, True, 'hardswish', 1, scale),
                  C(40, 5, 120, 48, True, 'hardswish', 1, scale), C(48, 5, 144, 48, True, 'hardswish', 1, scale),
                  C(48, 5, 288, 96, True, 'hardswish', 2, scale), C(96, 5, 576, 96, True, 'hardswish', 1, scale),
                  C(96, 5, 576, 96, True, 'hardswish', 1, scale)]                      # mobilenetv3.py:209-221
        super().__init__(config, last_channel=_make_divisible(1024 * scale, 8), scale=scale, with_pool=with_pool,
                         num_classes=num_classes)


class MobileNetV3Large(MobileNetV3):
    def __init__(self, scale=1.0, num_classes=1000, with_pool=True):
        C = InvertedResidualConfig
        config = [C(16, 3, 16, 16, False, 'relu', 1, scale), C(16, 3, 64, 24, False, 'relu', 2, scale),
                  C(24, 3, 72, 24, False, 'relu', 1, scale), C(24, 5, 72, 40, True, 'relu', 2, scale),
                  C(40, 5, 120, 40, True, 'relu', 1, scale), C(40, 5, 120, 40, True, 'relu', 1, scale),
                  C(40, 3, 240, 80, False, 'hardswish', 2, scale), C(80, 3, 200, 80, False, 'hardswish', 1, scale),
                  C(80, 3, 184, 80, False, 'hardswish', 1, scale), C(80, 3, 184, 80, False, 'hardswish', 1, scale),
                  C(80, 3, 480, 112, True, 'hardswish', 1, scale), C(112, 3, 672, 112, True, 'hardswish', 1, scale),
                  C(112, 5, 672, 160, True, 'hardswish', 2, scale), C(160, 5, 960, 160, True, 'hardswish', 1, scale),
                  C(160, 5, 960, 160, True, 'hardswish', 1, scale)]                    # mobilenetv3.py:253-269
        super().__init__(config, last_channel=_make_divisible(1280 * scale, 8), scale=scale, with_pool=with_pool,
                         num_classes=num_classes)


def mobilenet_v3_small(pretrained=False, scale=1.0, **kwargs):
    if pretrained:
        raise NotImplementedError("pretrained weights are not bundled; use model.load_weights(...)")
    return MobileNetV3Small(scale=scale, **kwargs)


def mobilenet_v3_large(pretrained=False, scale=1.0, **kwargs):
    if pretrained:
        raise NotImplementedError("pretrained weights are not bundled; use model.load_weights(...)")
    return MobileNetV3Large(scale=scale, **kwargs)
