"""ResNet-18/34/50/101/152 (+wide) forward graphs on the MI355X engine.

Same factory functions, constructor arguments, attribute names and parameter tree as the reference
(tlxcv/models/classification/resnet.py:16-382), so a weight dictionary keyed by attribute path
fits both.  What differs is the forward body: the reference issues conv, BatchNorm, ReLU and the
residual add as separate TensorLayerX layer calls (resnet.py:142-156, 286-300); here each
conv+BN(+ReLU)(+residual) is ONE implicit-GEMM launch with the folded BatchNorm, the skip
connection and the activation in its epilogue, tensors stay NHWC from the stem to the pool, and
the classifier is the same GEMM kernel with the bias as epilogue shift.
"""
import torch

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

    # -- the same block cut at the seams the engine fuses: run_head() = conv1 -> conv2 (+ projection shortcut) gives what
    #    conv3 needs; the caller then runs conv3 + skip + relu either alone (finish) or in ONE launch together with the
    #    next block's conv1 (E.bottleneck_seam): the wide map between two blocks is written once and not re-read.
    def run_head(self, v, t1=None):
        """v: the block's input; t1: relu(bn1(conv1(v))) if the previous seam launch already computed it.  -> conv2's output."""
        out = t1 if t1 is not None else self.conv1.run_nhwc(v, self.bn1, E.ACT_RELU)
        return self.conv2.run_nhwc(out, self.bn2, E.ACT_RELU)

    def shortcut(self, v):
        return v if self.downsample is None else self.downsample[0].run_nhwc(v, self.downsample[1])

    def finish(self, out, identity):
        return self.conv3.run_nhwc(out, self.bn3, E.ACT_RELU, res=identity)

    def seam_with(self, nxt, out, v):
        """(block output, nxt's conv1 output) in one launch — `v` is the block's input, the skip or the projection shortcut's
        source — or None when there is no fused kernel for these layers."""
        c3, c1 = self.conv3, nxt.conv1
        dt = E.precision()
        if (not E.option("seams") or dt != torch.float16 or c3.n_group != 1 or c1.n_group != 1 or c1.kernel_size != (1, 1) or c1.stride != (1, 1)
                or c1.padding != (0, 0) or c3.biases is not None or c1.biases is not None
                or not E.bottleneck_seam_supported(c3.in_channels, c3.out_channels, c1.out_channels, dt)
                or (c3.in_channels >= 256 and not E.option("seam256"))):
            return None
        # Where the fused launch pays (tools/small_batch.py, graph replay, one box): a seam runs its 64-channel steps one after
        # the other, so with few pixels it is a chain of latencies — batch 1 0.91 ms with all seams, 0.78 without the 14 x 14
        # ones, 0.74 without any; from 16 images the 56 x 56 / 28 x 28 seams win (batch 32: 1.09 vs 1.14 ms), the 14 x 14 ones
        # (1 MB of filters per 128 pixels) only from ~96 images a launch.
        n_img = out.shape[0]
        # (round 5, tools/batch_table.py: at 512 images = 2 x 256 per launch the 14 x 14 seams LOSE 2.3 % to two persistent GEMM launches —
        #  their 1 MB of filters per 128 pixels is re-streamed 196 times per CU; at 2 x 128 they win 1.7 %: fused up to 192 images a launch)
        #  end of round 5, same table: inside a two-stream forward of 96 / 128 images (48 / 64 a launch) they lose 1.1 / 1.9 % as well — 96 ... 192 images
        #  a launch whatever the stream arrangement)
        if n_img < 12 or (c3.in_channels >= 256 and (n_img < 96 or n_img > 192)):
            return None
        pk3 = c3._cached("pk", lambda: E.PackedFilter(c3.filters, dt))
        pk1 = c1._cached("pk", lambda: E.PackedFilter(c1.filters, dt))
        s3, h3 = c3._cached(("bn", id(self.bn3)), lambda: self.bn3.folded(None), deps=(self.bn3,))
        s1, h1 = c1._cached(("bn", id(nxt.bn1)), lambda: nxt.bn1.folded(None), deps=(nxt.bn1,))
        if self.downsample is None:
            return E.bottleneck_seam(out, pk3, s3, h3, v, pk1, s1, h1)
        cd, bnd = self.downsample[0], self.downsample[1]
        if (cd.kernel_size == (1, 1) and cd.stride == (1, 1) and cd.padding == (0, 0) and cd.n_group == 1 and cd.biases is None
                and c3.in_channels == 64 and c1.out_channels == 64 and cd.in_channels == 64 and v.shape[-1] == 64):
            # layer1.0: the projection shortcut (1x1, stride 1, 64 channels in) is computed inside the seam launch: its 411 MB
            # map (batch 256) is neither written nor read
            pkd = cd._cached("pk", lambda: E.PackedFilter(cd.filters, dt))
            sd, hd = cd._cached(("bn", id(bnd)), lambda: bnd.folded(None), deps=(bnd,))
            return E.bottleneck_seam(out, pk3, s3, h3, v, pk1, s1, h1, proj=(pkd, sd, hd))
        return E.bottleneck_seam(out, pk3, s3, h3, self.shortcut(v), pk1, s1, h1)


def run_bottleneck_chain(blocks, v):
    """A run of BottleneckBlocks (all stages of a ResNet in order) with every block-to-block seam fused where the library has the
    kernel (fp16; resnet.py:142-156 per block).  Falls back block by block to conv3 + skip as its own launch."""
    t1 = None
    for i, blk in enumerate(blocks):
        out = blk.run_head(v, t1)
        t1 = None
        nxt = blocks[i + 1] if i + 1 < len(blocks) else None
        fused = blk.seam_with(nxt, out, v) if isinstance(nxt, BottleneckBlock) else None
        if fused is not None:
            v, t1 = fused
        else:
            v = blk.finish(out, blk.shortcut(v))
    return v


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

    # tools/two_stream_threshold.py (round 5, ResNet-50; one stream / halves planned for half the CUs): batch 64 1.445 / 1.431 ms,
    # 80 1.531 / 1.516, 96 1.906 / 1.772, 128 2.153 / 2.037, 256 3.670 / 3.353
    @E.two_streams(96, plan="half")
    def forward(self, x):
        if (self.data_format == 'channels_first' and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
                and not x.permute(0, 2, 3, 1).is_contiguous()):
            # NCHW image: 2x2 space-to-depth fold + 4x4/1 conv == the 7x7/2 pad-3 stem (resnet.py:199-207, 287-289)
            # the max-pool (:290) rides in the stem's epilogue where the library has the fused kernel (224 x 224 inputs)
            v = self.conv1.run_stem(x, 2, self.bn1, E.ACT_RELU, maxpool=self.maxpool)
        else:
            v = self.maxpool.run_nhwc(self.conv1.run_nhwc(as_nhwc(x, self.data_format), self.bn1, E.ACT_RELU))   # :287-290
        blocks = [blk for layer in (self.layer1, self.layer2, self.layer3, self.layer4) for blk in layer]
        if all(isinstance(b, BottleneckBlock) for b in blocks):
            v = run_bottleneck_chain(blocks, v)                # seams between blocks fused (fp16)
        else:
            for blk in blocks:
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
