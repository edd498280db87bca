"""ResNet-18/34/50/101/152 (+wide) forward graphs on the MI355X engine.

Same factory functions, constructor arguments, attribute names and parameter tree as the reference
(tlxcv/models/classification/resnet.py:16-382), so a weight dictionary keyed by attribute path
fits both.  What differs is the forward body: the reference issues conv, BatchNorm, ReLU and the
residual add as separate TensorLayerX layer calls (resnet.py:142-156, 286-300); here each
conv+BN(+ReLU)(+residual) is ONE implicit-GEMM launch with the folded BatchNorm, the skip
connection and the activation in its epilogue, tensors stay NHWC from the stem to the pool, and
the classifier is the same GEMM kernel with the bias as epilogue shift.
"""
from ... import engine as E
from ...tlx import nn
from ...tlx import FlattenReshape
from ...tlx.nn import as_nhwc, from_nhwc

__all__ = ['resnet18', 'resnet34', 'resnet50', 'resnet101', 'resnet152', 'wide_resnet50_2', 'wide_resnet101_2',
           'ResNet', 'BasicBlock', 'BottleneckBlock']

_HE = nn.initializers.HeNormal()


def _conv(cin, cout, k, stride=1, padding=0, dilation=1, groups=1, data_format='channels_first'):
    return nn.GroupConv2d(in_channels=cin, out_channels=cout, kernel_size=k, stride=stride, padding=padding,
                          dilation=dilation, n_group=groups, b_init=(), W_init=_HE, data_format=data_format)


class BasicBlock(nn.Module):
    """resnet.py:16-77: 3x3 -> 3x3, BN after each, add, ReLU."""
    expansion = 1

    def __init__(self, in_channels, out_channels, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                 batch_norm=None, data_format='channels_first'):
        super().__init__()
        if batch_norm is None:
            batch_norm = nn.BatchNorm2d
        if dilation > 1:
            raise NotImplementedError('Dilation > 1 not supported in BasicBlock')
        self.conv1 = _conv(in_channels, out_channels, 3, stride, 1, data_format=data_format)
        self.bn1 = batch_norm(num_features=out_channels, data_format=data_format)
        self.relu = nn.ReLU()
        self.conv2 = _conv(out_channels, out_channels, 3, 1, 1, data_format=data_format)
        self.bn2 = batch_norm(num_features=out_channels, data_format=data_format)
        self.downsample = downsample
        self.stride = stride
        self.data_format = data_format

    def forward_nhwc(self, v):
        identity = v
        out = self.conv1.run_nhwc(v, self.bn1, E.ACT_RELU)
        if self.downsample is not None:
            identity = self.downsample[0].run_nhwc(v, self.downsample[1])
        return self.conv2.run_nhwc(out, self.bn2, E.ACT_RELU, res=identity)

    def forward(self, x):
        return from_nhwc(self.forward_nhwc(as_nhwc(x, self.data_format)), self.data_format)


class BottleneckBlock(nn.Module):
    """resnet.py:80-156: 1x1 -> 3x3 (stride here) -> 1x1 (x4), BN after each, add, ReLU."""
    expansion = 4

    def __init__(self, in_channels, out_channels, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                 batch_norm=None, data_format='channels_first'):
        super().__init__()
        if batch_norm is None:
            batch_norm = nn.BatchNorm2d
        width = int(out_channels * (base_width / 64.0)) * groups
        self.conv1 = _conv(in_channels, width, 1, data_format=data_format)
        self.bn1 = batch_norm(num_features=width, data_format=data_format)
        self.conv2 = _conv(width, width, 3, stride, dilation, dilation, groups, data_format=data_format)
        self.bn2 = batch_norm(num_features=width, data_format=data_format)
        self.conv3 = _conv(width, out_channels * self.expansion, 1, data_format=data_format)
        self.bn3 = batch_norm(num_features=out_channels * self.expansion, data_format=data_format)
        self.relu = nn.ReLU()
        self.downsample = downsample
        self.stride = stride
        self.data_format = data_format

    def forward_nhwc(self, v):
        identity = v
        out = self.conv1.run_nhwc(v, self.bn1, E.ACT_RELU)
        out = self.conv2.run_nhwc(out, self.bn2, E.ACT_RELU)
        if self.downsample is not None:
            identity = self.downsample[0].run_nhwc(v, self.downsample[1])
        # relu(bn3(conv3(out)) + identity): resnet.py:151-155, one launch
        return self.conv3.run_nhwc(out, self.bn3, E.ACT_RELU, res=identity)

    def forward(self, x):
        return from_nhwc(self.forward_nhwc(as_nhwc(x, self.data_format)), self.data_format)


class ResNet(nn.Module):
    """resnet.py:159-300."""

    def __init__(self, block, depth=50, width=64, num_classes=1000, with_pool=True, groups=1,
                 data_format='channels_first', name=None):
        super().__init__(name=name)
        layer_cfg = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}
        layers = layer_cfg[depth]
        self.groups = groups
        self.base_width = width
        self.num_classes = num_classes
        self.with_pool = with_pool
        self.in_channels = 64
        self.dilation = 1
        self.data_format = data_format
        self.conv1 = _conv(3, self.in_channels, 7, 2, 3, data_format=data_format)
        self.bn1 = nn.BatchNorm2d(num_features=self.in_channels, data_format=data_format)
        self.relu = nn.ReLU()
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1, data_format=data_format)
        self.layer1 = self._make_layer(block, 64, layers[0], data_format=data_format)
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2, data_format=data_format)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2, data_format=data_format)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2, data_format=data_format)
        if with_pool:
            self.avgpool = nn.AdaptiveAvgPool2d((1, 1), data_format=data_format)
        self.flatten = FlattenReshape()
        if num_classes > 0:
            self.fc = nn.Linear(in_features=512 * block.expansion, out_features=num_classes)

    def _make_layer(self, block, out_channels, blocks, stride=1, dilate=False, data_format='channels_first'):
        batch_norm = nn.BatchNorm2d
        downsample = None
        previous_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        if stride != 1 or self.in_channels != out_channels * block.expansion:
            downsample = nn.Sequential([
                _conv(self.in_channels, out_channels * block.expansion, 1, stride, data_format=data_format),
                batch_norm(num_features=out_channels * block.expansion, data_format=data_format)])
        layers = [block(self.in_channels, out_channels, stride, downsample, self.groups, self.base_width,
                        previous_dilation, batch_norm, data_format=data_format)]
        self.in_channels = out_channels * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.in_channels, out_channels, groups=self.groups, base_width=self.base_width,
                                batch_norm=batch_norm, data_format=data_format))
        return nn.Sequential(layers)

    def forward(self, x):
        if (self.data_format == 'channels_first' and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
                and not x.permute(0, 2, 3, 1).is_contiguous()):
            # NCHW image: 2x2 space-to-depth fold + 4x4/1 conv == the 7x7/2 pad-3 stem (resnet.py:199-207, 287-289)
            # the max-pool (:290) rides in the stem's epilogue where the library has the fused kernel (224 x 224 inputs)
            v = self.conv1.run_stem(x, 2, self.bn1, E.ACT_RELU, maxpool=self.maxpool)
        else:
            v = self.maxpool.run_nhwc(self.conv1.run_nhwc(as_nhwc(x, self.data_format), self.bn1, E.ACT_RELU))   # :287-290
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                v = blk.forward_nhwc(v)
        if self.with_pool:
            v = E.global_avgpool(v)                            # (N, C)   :295-296
            if self.num_classes > 0:
                return self.fc.run(v)                          # flatten is a no-op on (N, C)   :297-299
            N, Cc = v.shape
            return v.view(N, 1, 1, Cc) if self.data_format == 'channels_last' else v.view(N, Cc, 1, 1)
        y = from_nhwc(v, self.data_format)
        if self.num_classes > 0:
            return self.fc(self.flatten(y.contiguous()))
        return y


def _resnet(arch, Block, depth, pretrained, **kwargs):
    if pretrained:
        raise NotImplementedError("pretrained weights are not bundled; use model.load_weights(...)")
    return ResNet(Block, depth, name=arch, **kwargs)


def resnet18(pretrained=False, **kwargs):
    return _resnet('resnet18', BasicBlock, 18, pretrained, **kwargs)


def resnet34(pretrained=False, **kwargs):
    return _resnet('resnet34', BasicBlock, 34, pretrained, **kwargs)


def resnet50(pretrained=False, **kwargs):
    return _resnet('resnet50', BottleneckBlock, 50, pretrained, **kwargs)


def resnet101(pretrained=False, **kwargs):
    return _resnet('resnet101', BottleneckBlock, 101, pretrained, **kwargs)


def resnet152(pretrained=False, **kwargs):
    return _resnet('resnet152', BottleneckBlock, 152, pretrained, **kwargs)


def wide_resnet50_2(pretrained=False, **kwargs):
    return _resnet('wide_resnet50_2', BottleneckBlock, 50, pretrained, width=128, **kwargs)


def wide_resnet101_2(pretrained=False, **kwargs):
    return _resnet('wide_resnet101_2', BottleneckBlock, 101, pretrained, width=128, **kwargs)
