"""EfficientNet-B0..B7 forward graph on the MI355X engine — same constructors / parameter tree as
tlxcv/models/classification/efficientnet.py:13-547 (`features.<i>.<j>.block.<k>.0.filters/biases`, `.1.gamma...`,
`.fc1/.fc2.filters/biases`, `classifier.1.weights/biases`).

The reference builds every layer without input channels and runs one forward of ones at construction to create the
weights (:433-441); here the channel counts are derived from the same MBConvConfig arithmetic (:197-226), so the
parameter tree exists — with the same names and shapes — before any tensor is seen.

Fusions: ConvNormActivation (:74-125: GroupConv2d 'SAME' with bias + BatchNorm2d + SiLU) is ONE launch — the implicit
GEMM for 1x1 / 3x3 dense convs, the depthwise kernel for the k x k depthwise convs, bias folded into the BatchNorm
shift, SiLU in the epilogue; 'SAME' at stride 2 is the one-sided end padding of tlxmi_conv2d / tlxmi_dwconv2d;
SqueezeExcitation (:128-178) = global-average-pool kernel + two tiny GEMMs (SiLU / sigmoid epilogues) + the
channel-scale kernel; the residual of MBConv (:302-307, StochasticDepth is the identity in eval) rides in the project
conv's epilogue; avgpool + Flatten + Dropout + Linear (:428-431) = pool kernel + one GEMM."""
import copy
import math
from functools import partial

import torch

from ... import engine as E
from ...tlx import nn
from ...tlx.nn import as_nhwc, from_nhwc

__all__ = ["efficientnet", "EfficientNet"]


class SiLU(nn.Module):
    ACT = E.ACT_SILU

    def forward(self, x):
        v = as_nhwc(x, 'channels_first') if x.dim() == 4 else x
        y = E.affine_act(v.contiguous(), act=E.ACT_SILU)
        return from_nhwc(y, 'channels_first') if x.dim() == 4 else y


class StochasticDepth(nn.Module):
    """efficientnet.py:21-71: active in training only."""

    def __init__(self, p, mode):
        super().__init__()
        self.p, self.mode = p, mode

    def forward(self, input):
        self._require_eval()
        return input


def _make_divisible(v, divisor, min_value=None):
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


class ConvNormActivation(nn.Sequential):
    def __init__(self, out_channels, kernel_size=3, stride=1, padding=None, groups=1, norm_layer=nn.BatchNorm2d,
                 activation_layer=nn.ReLU, dilation=1, data_format="channels_first", in_channels=None):
        layers = [nn.GroupConv2d(out_channels, (kernel_size, kernel_size), (stride, stride), groups, None, "SAME",
                                 dilation=(dilation, dilation), W_init="he_normal", b_init="zeros",
                                 data_format=data_format, in_channels=in_channels)]
        if norm_layer is not None:
            layers.append(norm_layer(num_features=out_channels, data_format=data_format))
        if activation_layer is not None:
            layers.append(activation_layer())
        super().__init__(*layers)
        self.out_channels = out_channels
        self.data_format = data_format

    def run_nhwc(self, v, res=None):
        mods = list(self)
        bn = mods[1] if len(mods) > 1 and isinstance(mods[1], nn.BatchNorm2d) else None
        act = mods[-1].ACT if hasattr(mods[-1], "ACT") else E.ACT_NONE
        if res is not None:     # depthwise convs never carry the residual here (MBConv adds it after the project conv)
            return mods[0].run_nhwc(v, bn, act, res=res)
        return mods[0].run_nhwc(v, bn, act)

    def forward(self, x):
        return from_nhwc(self.run_nhwc(as_nhwc(x, self.data_format)), self.data_format)


class SqueezeExcitation(nn.Module):
    def __init__(self, input_channels, squeeze_channels, activation=nn.ReLU, scale_activation=nn.Sigmoid,
                 data_format="channels_first"):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d(1, data_format=data_format)
        self.fc1 = nn.Conv2d(squeeze_channels, (1, 1), padding="VALID", W_init="he_normal", b_init="zeros",
                             data_format=data_format, in_channels=input_channels)
        self.fc2 = nn.Conv2d(input_channels, (1, 1), padding="VALID", W_init="he_normal", b_init="zeros",
                             data_format=data_format, in_channels=squeeze_channels)
        self.activation = activation()
        self.scale_activation = scale_activation()
        self.squeeze_channels = squeeze_channels
        self.data_format = data_format

    def run_nhwc(self, v):
        N, Cc = v.shape[0], v.shape[-1]
        s = E.global_avgpool(v).view(N, 1, 1, Cc)
        # squeeze widths (max(1, in // 4): 4, 6, 10, ...) are not whole 16-byte chunks: fc1 writes into a zeroed buffer
        # of the padded width, which is what fc2's packed filter (input channels padded with zeros) reads
        vec = E.vec(v.dtype)
        sq_pad = (self.squeeze_channels + vec - 1) // vec * vec
        mid = torch.zeros((N, 1, 1, sq_pad), dtype=v.dtype, device=v.device)
        self.fc1.run_nhwc(s, act=self.activation.ACT, out=mid, out_ld=sq_pad)
        s = self.fc2.run_nhwc(mid, act=self.scale_activation.ACT)
        return E.scale_channels(v, s.view(N, Cc))

    def forward(self, input):
        return from_nhwc(self.run_nhwc(as_nhwc(input, self.data_format)), self.data_format)


class MBConvConfig:
    def __init__(self, expand_ratio, kernel, stride, input_channels, out_channels, num_layers, width_mult, depth_mult):
        self.expand_ratio = expand_ratio
        self.kernel = kernel
        self.stride = stride
        self.input_channels = self.adjust_channels(input_channels, width_mult)
        self.out_channels = self.adjust_channels(out_channels, width_mult)
        self.num_layers = self.adjust_depth(num_layers, depth_mult)

    @staticmethod
    def adjust_channels(channels, width_mult, min_value=None):
        return _make_divisible(channels * width_mult, 8, min_value)

    @staticmethod
    def adjust_depth(num_layers, depth_mult):
        return int(math.ceil(num_layers * depth_mult))


class MBConv(nn.Module):
    def __init__(self, cnf, stochastic_depth_prob, norm_layer, se_layer=SqueezeExcitation, data_format="channels_first"):
        super().__init__()
        if not (1 <= cnf.stride <= 2):
            raise ValueError("illegal stride value")
        self.use_res_connect = cnf.stride == 1 and cnf.input_channels == cnf.out_channels
        layers = []
        expanded_channels = cnf.adjust_channels(cnf.input_channels, cnf.expand_ratio)
        if expanded_channels != cnf.input_channels:
            layers.append(ConvNormActivation(expanded_channels, kernel_size=1, norm_layer=norm_layer, activation_layer=SiLU,
                                             data_format=data_format, in_channels=cnf.input_channels))
        layers.append(ConvNormActivation(expanded_channels, kernel_size=cnf.kernel, stride=cnf.stride,
                                         groups=expanded_channels, norm_layer=norm_layer, activation_layer=SiLU,
                                         data_format=data_format, in_channels=expanded_channels))
        squeeze_channels = max(1, cnf.input_channels // 4)
        layers.append(se_layer(expanded_channels, squeeze_channels, activation=SiLU, data_format=data_format))
        layers.append(ConvNormActivation(cnf.out_channels, kernel_size=1, norm_layer=norm_layer, activation_layer=None,
                                         data_format=data_format, in_channels=expanded_channels))
        self.block = nn.Sequential(*layers)
        self.stochastic_depth = StochasticDepth(stochastic_depth_prob, "row")
        self.out_channels = cnf.out_channels
        self.data_format = data_format

    def forward_nhwc(self, v):
        mods = list(self.block)
        y = v
        for m in mods[:-1]:
            y = m.run_nhwc(y)
        return mods[-1].run_nhwc(y, res=v if self.use_res_connect else None)      # :303-306

    def forward(self, input):
        return from_nhwc(self.forward_nhwc(as_nhwc(input, self.data_format)), self.data_format)


class EfficientNet(nn.Module):
    def __init__(self, inverted_residual_setting, dropout, stochastic_depth_prob=0.2, num_labels=1000, block=None,
                 norm_layer=None, input_shape=224, data_format="channels_first", **kwargs):
        super().__init__()
        if not inverted_residual_setting:
            raise ValueError("The inverted_residual_setting should not be empty")
        elif not (isinstance(inverted_residual_setting, (list, tuple))
                  and all(isinstance(s, MBConvConfig) for s in inverted_residual_setting)):
            raise TypeError("The inverted_residual_setting should be List[MBConvConfig]")
        if block is None:
            block = MBConv
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        self.data_format = data_format
        layers = []
        firstconv_output_channels = inverted_residual_setting[0].input_channels
        layers.append(ConvNormActivation(firstconv_output_channels, kernel_size=3, stride=2, norm_layer=norm_layer,
                                         activation_layer=SiLU, data_format=data_format, in_channels=3))
        total_stage_blocks = sum(cnf.num_layers for cnf in inverted_residual_setting)
        stage_block_id = 0
        for cnf in inverted_residual_setting:
            stage = []
            for _ in range(cnf.num_layers):
                block_cnf = copy.copy(cnf)
                if stage:
                    block_cnf.input_channels = block_cnf.out_channels
                    block_cnf.stride = 1
                sd_prob = stochastic_depth_prob * float(stage_block_id) / total_stage_blocks
                stage.append(block(block_cnf, sd_prob, norm_layer, data_format=data_format))
                stage_block_id += 1
            layers.append(nn.Sequential(stage))
        lastconv_input_channels = inverted_residual_setting[-1].out_channels
        lastconv_output_channels = 4 * lastconv_input_channels
        layers.append(ConvNormActivation(lastconv_output_channels, kernel_size=1, norm_layer=norm_layer,
                                         activation_layer=SiLU, data_format=data_format,
                                         in_channels=lastconv_input_channels))
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d(1, data_format=data_format)
        self.flatten = nn.Flatten()
        self.classifier = nn.Sequential(
            nn.Dropout(p=dropout),
            nn.Linear(num_labels, W_init=nn.initializers.random_uniform(-1.0 / math.sqrt(num_labels), 1.0 / math.sqrt(num_labels)),
                      b_init="zeros", in_features=lastconv_output_channels),
        )

    @E.two_streams(128, plan="full", eager=False)      # ~250 tiny launches, host-bound kernel by kernel
    def forward(self, x):
        feats = list(self.features)
        first = list(feats[0])
        if (self.data_format == "channels_first" and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
                and not x.permute(0, 2, 3, 1).is_contiguous()):
            # 3x3 / 2 'SAME' stem on the 2x2 space-to-depth image: 2x2 taps over 12 channels, K dense (:354-363)
            v = first[0].run_stem(x, 2, first[1], first[2].ACT)
        else:
            # the RGB image with channels padded to a whole 16-byte chunk (the packed filter pads with zeros)
            v = feats[0].run_nhwc(E.nchw_to_nhwc(x if self.data_format == "channels_first" else x.permute(0, 3, 1, 2), E.precision()))
        for m in feats[1:]:
            if isinstance(m, ConvNormActivation):
                v = m.run_nhwc(v)
            else:
                for blk in m:
                    v = blk.forward_nhwc(v)
        v = E.global_avgpool(v)                                    # :428-429
        return list(self.classifier)[1].run(v)                     # :430 (Dropout = identity in eval)


def _efficientnet(width_mult, depth_mult, dropout, input_shape, data_format="channels_first", **kwargs):
    inverted_residual_setting = [
        MBConvConfig(1, 3, 1, 32, 16, 1, width_mult, depth_mult),
        MBConvConfig(6, 3, 2, 16, 24, 2, width_mult, depth_mult),
        MBConvConfig(6, 5, 2, 24, 40, 2, width_mult, depth_mult),
        MBConvConfig(6, 3, 2, 40, 80, 3, width_mult, depth_mult),
        MBConvConfig(6, 5, 1, 80, 112, 3, width_mult, depth_mult),
        MBConvConfig(6, 5, 2, 112, 192, 4, width_mult, depth_mult),
        MBConvConfig(6, 3, 1, 192, 320, 1, width_mult, depth_mult),
    ]
    return EfficientNet(inverted_residual_setting, dropout, input_shape=input_shape, data_format=data_format, **kwargs)


_ARCH = {   # efficientnet.py:465-547: width, depth, dropout, input size, BatchNorm (epsilon, momentum)
    "efficientnet_b0": (1.0, 1.0, 0.2, 224, 1e-5, 0.1), "efficientnet_b1": (1.0, 1.1, 0.2, 240, 1e-5, 0.1),
    "efficientnet_b2": (1.1, 1.2, 0.3, 260, 1e-5, 0.1), "efficientnet_b3": (1.2, 1.4, 0.3, 300, 1e-5, 0.1),
    "efficientnet_b4": (1.4, 1.8, 0.4, 380, 1e-5, 0.1), "efficientnet_b5": (1.6, 2.2, 0.4, 456, 0.001, 0.01),
    "efficientnet_b6": (1.8, 2.6, 0.5, 528, 0.001, 0.01), "efficientnet_b7": (2.0, 3.1, 0.5, 600, 0.001, 0.01),
}


def efficientnet(arch, data_format="channels_first", **kwargs):
    if arch not in _ARCH:
        raise ValueError(f"unknown EfficientNet variant {arch!r}")
    w, d, drop, size, eps, mom = _ARCH[arch]
    return _efficientnet(w, d, drop, size, norm_layer=partial(nn.BatchNorm2d, epsilon=eps, momentum=mom),
                         data_format=data_format, **kwargs)
