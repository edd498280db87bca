"""MobileNetV1 forward graph on the MI355X engine — same constructor / parameter tree as
tlxcv/models/classification/mobilenetv1.py:7-262.  Every Conv-BN-ReLU triple (:45-65) is one launch:
the 3x3 depthwise convs run the HBM-bound tlxmi_dwconv2d kernel (no MFMA: 9 MACs per element), the
1x1 pointwise convs the implicit-GEMM kernel, both with the folded BatchNorm + ReLU epilogue."""
from ... import engine as E
from ... import tlx
from ...tlx import nn
from ...tlx.nn import as_nhwc, from_nhwc

__all__ = ["MobileNetV1"]

_ACT_CODE = {nn.ReLU: E.ACT_RELU, nn.ReLU6: E.ACT_RELU6, nn.Hardswish: E.ACT_HARDSWISH, nn.Sigmoid: E.ACT_SIGMOID}


class ConvNormActivation(nn.Sequential):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=None, groups=1,
                 batch_norm=nn.BatchNorm2d, activation_layer=nn.ReLU, dilation=1, bias=None,
                 data_format="channels_first"):
        if padding is None:
            padding = (kernel_size - 1) // 2 * dilation
        if bias is None:
            bias = batch_norm is None
        layers = [nn.GroupConv2d(dilation=dilation, in_channels=in_channels, out_channels=out_channels,
                                 kernel_size=kernel_size, stride=stride, padding=padding, b_init=bias, n_group=groups,
                                 W_init=nn.initializers.HeNormal(), data_format=data_format)]
        if batch_norm is not None:
            layers.append(batch_norm(num_features=out_channels, data_format=data_format))
        if activation_layer is not None:
            layers.append(activation_layer())
        super().__init__(*layers)
        self._has_bn = batch_norm is not None
        self._act = _ACT_CODE.get(activation_layer, None) if activation_layer is not None else E.ACT_NONE
        self.data_format = data_format

    def run_nhwc(self, v, res=None):
        mods = list(self)
        bn = mods[1] if self._has_bn else None
        if self._act is None:  # an activation the epilogue does not know: run it as its own layer
            y = mods[0].run_nhwc(v, bn, res=res)
            return as_nhwc(mods[-1](from_nhwc(y, 'channels_first')), 'channels_first')
        return mods[0].run_nhwc(v, bn, self._act, res=res)

    def forward(self, x):
        return from_nhwc(self.run_nhwc(as_nhwc(x, self.data_format)), self.data_format)


class DepthwiseSeparable(nn.Module):
    def __init__(self, in_channels, out_channels1, out_channels2, num_groups, stride, scale,
                 data_format="channels_first", name=None):
        super().__init__(name=name)
        self._depthwise_conv = ConvNormActivation(in_channels, int(out_channels1 * scale), kernel_size=3, stride=stride,
                                                  padding=1, groups=int(num_groups * scale), data_format=data_format)
        self._pointwise_conv = ConvNormActivation(int(out_channels1 * scale), int(out_channels2 * scale), kernel_size=1,
                                                  stride=1, padding=0, data_format=data_format)
        self.data_format = data_format

    def run_nhwc(self, v):
        return self._pointwise_conv.run_nhwc(self._depthwise_conv.run_nhwc(v))

    def forward(self, x):
        return from_nhwc(self.run_nhwc(as_nhwc(x, self.data_format)), self.data_format)


class MobileNetV1(nn.Module):
    def __init__(self, scale=1.0, num_classes=1000, with_pool=True, data_format="channels_first"):
        super().__init__()
        self.scale, self.num_classes, self.with_pool, self.data_format = scale, num_classes, with_pool, data_format
        self.conv1 = ConvNormActivation(in_channels=3, out_channels=int(32 * scale), kernel_size=3, stride=2, padding=1,
                                        data_format=data_format)
        plan = [(32, 32, 64, 32, 1), (64, 64, 128, 64, 2), (128, 128, 128, 128, 1), (128, 128, 256, 128, 2),
                (256, 256, 256, 256, 1), (256, 256, 512, 256, 2)] + [(512, 512, 512, 512, 1)] * 5 + \
               [(512, 512, 1024, 512, 2), (1024, 1024, 1024, 1024, 1)]          # mobilenetv1.py:136-244
        self.dwsl = nn.Sequential(*[
            DepthwiseSeparable(in_channels=int(cin * scale), out_channels1=c1, out_channels2=c2, num_groups=g,
                               stride=s, scale=scale, data_format=data_format) for cin, c1, c2, g, s in plan])
        if with_pool:
            self.pool2d_avg = nn.AdaptiveAvgPool2d(1, data_format=data_format)
        if num_classes > 0:
            self.fc = nn.Linear(in_features=int(1024 * scale), out_features=num_classes)

    @E.two_streams(128, plan="full")
    def forward(self, x):
        v = self.conv1.run_nhwc(as_nhwc(x, self.data_format))      # :255
        for blk in self.dwsl:                                      # :256
            v = blk.run_nhwc(v)
        if self.with_pool:
            v = E.global_avgpool(v)                                # :258
            if self.num_classes > 0:
                return self.fc.run(v)                              # :260-261
            return v.view(v.shape[0], -1, 1, 1)
        y = from_nhwc(v, self.data_format)
        if self.num_classes > 0:
            return self.fc(tlx.reshape(y.contiguous(), (y.shape[0], -1)))
        return y
