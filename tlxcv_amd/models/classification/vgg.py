"""VGG-11/13/16/19 (with or without BatchNorm) forward graph on the MI355X engine — same constructor /
parameter tree as tlxcv/models/classification/vgg.py:9-159 (`features.<i>.filters/biases`, BatchNorm
`features.<i>.gamma...`, `classifier.{0,3,6}.weights/biases`).

Fusions: every Conv(3x3, padding='SAME', bias)(+BatchNorm)+ReLU run of `make_layers` (:61-90) is ONE
implicit-GEMM launch (bias / folded BatchNorm + ReLU in the epilogue); MaxPool2d(2,2) is the pooling
kernel; AdaptiveAvgPool2d((7,7)) (:36-39) is the identity at 224 x 224 and a windowed mean otherwise;
FlattenReshape (:40,56) flattens in the reference's (C, H, W) order, so the NHWC map is turned once
(12.8 MB at batch 256); the classifier (:42-50) is three Linear launches with bias + ReLU epilogues
(Dropout is the identity in eval)."""
from ... import engine as E
from ...tlx import nn
from ...tlx.nn import as_nhwc, from_nhwc

__all__ = ['VGG', 'vgg11', 'vgg13', 'vgg16', 'vgg19']


class VGG(nn.Module):
    def __init__(self, features, num_classes=1000, with_pool=True, data_format='channels_first', name=None):
        super().__init__(name)
        self.features = features
        self.num_classes = num_classes
        self.with_pool = with_pool
        self.data_format = data_format
        if with_pool:
            self.avgpool = nn.AdaptiveAvgPool2d((7, 7), data_format=data_format)
        if num_classes > 0:
            self.classifier = nn.Sequential([
                nn.Linear(in_features=25088, out_features=4096),
                nn.ReLU(),
                nn.Dropout(),
                nn.Linear(in_features=4096, out_features=4096),
                nn.ReLU(),
                nn.Dropout(),
                nn.Linear(in_features=4096, out_features=num_classes)
            ])

    def features_nhwc(self, v):
        """vgg.py:53 — conv(+bn)+relu runs fused, pools as they come."""
        mods = list(self.features)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.GroupConv2d):
                bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d) else None
                j = i + (2 if bn is not None else 1)
                relu = j < len(mods) and isinstance(mods[j], nn.ReLU)
                v = m.run_nhwc(v, bn, E.ACT_RELU if relu else E.ACT_NONE)
                i = j + (1 if relu else 0)
            elif isinstance(m, nn.MaxPool2d):
                v = E.maxpool2d(v, m.kernel_size, m.stride, m.padding)
                i += 1
            else:
                v = as_nhwc(m(from_nhwc(v, 'channels_first')), 'channels_first')
                i += 1
        return v

    @E.two_streams(32, plan="half")
    def forward(self, x):
        v = self.features_nhwc(as_nhwc(x, self.data_format))
        if self.with_pool:
            v = self.avgpool.run_nhwc(v)                                   # :54-55
        if self.num_classes > 0:
            N, H, W, C = v.shape
            flat = E.nhwc_to_nchw(v).view(N, C * H * W)                    # FlattenReshape on NCHW, :56
            c = list(self.classifier)
            h = c[0].run(flat, act=E.ACT_RELU)                             # :57, Linear + ReLU (+ Dropout = id)
            h = c[3].run(h, act=E.ACT_RELU)
            return c[6].run(h)
        return from_nhwc(v, self.data_format)


def make_layers(cfg, batch_norm=False, data_format='channels_first'):
    layers = []
    in_channels = 3
    for v in cfg:
        if v == 'M':
            layers += [nn.MaxPool2d(kernel_size=2, stride=2, data_format=data_format)]
        else:
            conv2d = nn.GroupConv2d(kernel_size=3, padding='SAME', in_channels=in_channels, out_channels=v,
                                    data_format=data_format)
            if batch_norm:
                layers += [conv2d, nn.BatchNorm2d(num_features=v, data_format=data_format), nn.ReLU()]
            else:
                layers += [conv2d, nn.ReLU()]
            in_channels = v
    return nn.Sequential(layers)


cfgs = {    # vgg.py:93-98
    'A': [64, 'M', 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M'],
    'B': [64, 64, 'M', 128, 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M'],
    'D': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M'],
    'E': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']
}


def _vgg(name, cfg, batch_norm, data_format, **kwargs):
    features = make_layers(cfgs[cfg], batch_norm, data_format)
    return VGG(features, data_format=data_format, name=name, **kwargs)


def vgg11(batch_norm=False, data_format='channels_first', **kwargs):
    return _vgg('vgg11_bn' if batch_norm else 'vgg11', 'A', batch_norm, data_format, **kwargs)


def vgg13(batch_norm=False, data_format='channels_first', **kwargs):
    return _vgg('vgg13_bn' if batch_norm else 'vgg13', 'B', batch_norm, data_format, **kwargs)


def vgg16(batch_norm=False, data_format='channels_first', **kwargs):
    return _vgg('vgg16_bn' if batch_norm else 'vgg16', 'D', batch_norm, data_format, **kwargs)


def vgg19(batch_norm=False, data_format='channels_first', **kwargs):
    return _vgg('vgg19_bn' if batch_norm else 'vgg19', 'E', batch_norm, data_format, **kwargs)
