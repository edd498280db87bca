"""AlexNet forward graph on the MI355X engine — same constructor / parameter tree as
tlxcv/models/classification/alexnet.py:11-181 (`_conv1._conv.filters`, ..., `_fc6.weights`).

Fusions: conv + bias + ReLU of ConvPoolLayer.forward (:44-49) and of `_conv3` / `_conv4` + tlx.relu
(:155-158) are one implicit-GEMM launch each; the 11x11 stride-4 stem runs on a 4x4 space-to-depth input
(4x4 taps over 48 folded channels instead of 121 taps over 3-of-8 channels); MaxPool2d(3,2,0) is the pooling
kernel; tlx.flatten (:161) is the reference's (C, H, W) order, so the 6x6x256 map is turned once; fc6 / fc7 carry
bias + ReLU in their epilogues (:163-168), Dropout is the identity in eval."""
import math

from ... import engine as E
from ...tlx import nn
from ...tlx.nn import GroupConv2d, Linear, ReLU, as_nhwc, from_nhwc
from ...tlx.nn.initializers import random_uniform, xavier_uniform

__all__ = ["AlexNet", "alexnet"]


class ConvPoolLayer(nn.Module):
    def __init__(self, input_channels, output_channels, filter_size, stride, padding, stdv, groups=1, act=None,
                 data_format='channels_first'):
        super().__init__()
        self.relu = ReLU() if act == 'relu' else None
        self._conv = GroupConv2d(in_channels=input_channels, out_channels=output_channels, kernel_size=filter_size,
                                 stride=stride, padding=padding, W_init=random_uniform(), b_init=random_uniform(),
                                 n_group=groups, data_format=data_format)
        self._pool = nn.MaxPool2d(kernel_size=3, stride=2, padding=0, data_format=data_format)
        self.data_format = data_format

    def run_nhwc(self, v):
        y = self._conv.run_nhwc(v, None, E.ACT_RELU if self.relu is not None else E.ACT_NONE)
        return E.maxpool2d(y, self._pool.kernel_size, self._pool.stride, self._pool.padding)

    def run_stem(self, x_nchw, fold):
        y = self._conv.run_stem(x_nchw, fold, None, E.ACT_RELU if self.relu is not None else E.ACT_NONE)
        return E.maxpool2d(y, self._pool.kernel_size, self._pool.stride, self._pool.padding)

    def forward(self, inputs):
        return from_nhwc(self.run_nhwc(as_nhwc(inputs, self.data_format)), self.data_format)


class AlexNet(nn.Module):
    def __init__(self, num_classes=1000, data_format='channels_first', name=None):
        super().__init__(name)
        self.num_classes = num_classes
        self.data_format = data_format
        stdv = 1.0 / math.sqrt(3 * 11 * 11)
        self._conv1 = ConvPoolLayer(3, 64, 11, 4, 2, stdv, act='relu', data_format=data_format)
        stdv = 1.0 / math.sqrt(64 * 5 * 5)
        self._conv2 = ConvPoolLayer(64, 192, 5, 1, 2, stdv, act='relu', data_format=data_format)
        self._conv3 = GroupConv2d(stride=1, padding=1, in_channels=192, out_channels=384, kernel_size=3,
                                  W_init=random_uniform(), b_init=random_uniform(), data_format=data_format)
        self._conv4 = GroupConv2d(stride=1, padding=1, in_channels=384, out_channels=256, kernel_size=3,
                                  W_init=random_uniform(), b_init=random_uniform(), data_format=data_format)
        stdv = 1.0 / math.sqrt(256 * 3 * 3)
        self._conv5 = ConvPoolLayer(256, 256, 3, 1, 1, stdv, act='relu', data_format=data_format)
        if self.num_classes > 0:
            stdv = 1.0 / math.sqrt(256 * 6 * 6)
            self._drop1 = nn.Dropout(p=0.5)
            self._fc6 = Linear(in_features=9216, out_features=4096, W_init=random_uniform(-stdv, stdv),
                               b_init=xavier_uniform())
            self._drop2 = nn.Dropout(p=0.5)
            self._fc7 = Linear(in_features=4096, out_features=4096, W_init=random_uniform(-stdv, stdv),
                               b_init=xavier_uniform())
            self._fc8 = Linear(in_features=4096, out_features=num_classes, W_init=random_uniform(-stdv, stdv),
                               b_init=xavier_uniform())

    @E.two_streams(128, plan="full")
    def forward(self, inputs):
        E.need_gpu(inputs, "input")
        H, W = (inputs.shape[2], inputs.shape[3]) if self.data_format == 'channels_first' else (inputs.shape[1], inputs.shape[2])
        if self.data_format == 'channels_first' and H % 4 == 0 and W % 4 == 0:
            v = self._conv1.run_stem(inputs, 4)                            # :153, 11x11/4 on a 4x4 fold
        else:
            v = self._conv1.run_nhwc(as_nhwc(inputs, self.data_format))
        v = self._conv2.run_nhwc(v)                                        # :154
        v = self._conv3.run_nhwc(v, None, E.ACT_RELU)                      # :155-156
        v = self._conv4.run_nhwc(v, None, E.ACT_RELU)                      # :157-158
        v = self._conv5.run_nhwc(v)                                        # :159
        if self.num_classes > 0:
            N, Hh, Ww, C = v.shape
            flat = E.nhwc_to_nchw(v).view(N, C * Hh * Ww)                  # tlx.flatten on NCHW, :161
            h = self._fc6.run(flat, act=E.ACT_RELU)                        # :163-164
            h = self._fc7.run(h, act=E.ACT_RELU)                           # :166-167
            return self._fc8.run(h)                                        # :168
        return from_nhwc(v, self.data_format)


def _alexnet(arch, pretrained, **kwargs):
    return AlexNet(**kwargs)


def alexnet(pretrained=False, **kwargs):
    return _alexnet('alexnet', pretrained, **kwargs)
