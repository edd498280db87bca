"""DarkNet-53 forward graph on the MI355X engine — same classes / parameter tree as
tlxcv/models/detection/backbones/darknet.py:7-312.  ConvBNLayer (conv, BatchNorm, LeakyReLU(0.1), :54-58)
is one implicit-GEMM launch; BasicBlock's `tlx.add(value=inputs, bias=conv2)` (:155-159) rides in the
second conv's epilogue as a residual added AFTER the activation."""
from .... import engine as E
from ....tlx import nn
from ....tlx.nn import as_nhwc, from_nhwc

__all__ = ["DarkNet", "ConvBNLayer"]


class ConvBNLayer(nn.Module):
    def __init__(self, ch_in, ch_out, filter_size=3, stride=1, groups=1, padding=0, act="leaky",
                 data_format="channels_first", name="", **kwargs):
        super().__init__(name=name)
        self.conv = nn.GroupConv2d(in_channels=ch_in, out_channels=ch_out, kernel_size=filter_size, stride=stride,
                                   padding=padding, data_format=data_format, b_init=False, n_group=groups,
                                   W_init=nn.initializers.HeNormal())
        self.batch_norm = nn.BatchNorm2d(num_features=ch_out, data_format=data_format)
        if act == "leaky":
            self.act = nn.LeakyReLU(0.1)
        else:
            raise NotImplementedError
        self.data_format = data_format

    def run_nhwc(self, v, res=None, **kw):
        return self.conv.run_nhwc(v, self.batch_norm, E.ACT_LEAKY, 0.1, res=res, res_after_act=res is not None, **kw)

    def forward(self, inputs):
        return from_nhwc(self.run_nhwc(as_nhwc(inputs, self.data_format)), self.data_format)


class DownSample(nn.Module):
    def __init__(self, ch_in, ch_out, filter_size=3, stride=2, padding=1, norm_type="bn", norm_decay=0.0,
                 freeze_norm=False, data_format="channels_first"):
        super().__init__()
        self.conv_bn_layer = ConvBNLayer(ch_in=ch_in, ch_out=ch_out, filter_size=filter_size, stride=stride,
                                         padding=padding, data_format=data_format)
        self.ch_out = ch_out

    def run_nhwc(self, v):
        return self.conv_bn_layer.run_nhwc(v)

    def forward(self, inputs):
        return self.conv_bn_layer(inputs)


class BasicBlock(nn.Module):
    def __init__(self, ch_in, ch_out, norm_type="bn", norm_decay=0.0, freeze_norm=False, data_format="channels_first"):
        super().__init__()
        assert ch_in == ch_out and ch_in % 2 == 0, \
            f"ch_in and ch_out should be the same even int, but the input 'ch_in is {ch_in}, 'ch_out is {ch_out}"
        self.conv1 = ConvBNLayer(ch_in=ch_in, ch_out=int(ch_out / 2), filter_size=1, stride=1, padding=0,
                                 data_format=data_format)
        self.conv2 = ConvBNLayer(ch_in=int(ch_out / 2), ch_out=ch_out, filter_size=3, stride=1, padding=1,
                                 data_format=data_format)
        self.data_format = data_format

    def run_nhwc(self, v):
        return self.conv2.run_nhwc(self.conv1.run_nhwc(v), res=v)       # inputs + leaky(bn(conv2(...)))

    def forward(self, inputs):
        return from_nhwc(self.run_nhwc(as_nhwc(inputs, self.data_format)), self.data_format)


class Blocks(nn.Module):
    def __init__(self, ch_in, ch_out, count, norm_type="bn", norm_decay=0.0, freeze_norm=False, name=None,
                 data_format="channels_first"):
        super().__init__(name=name)
        self.basicblock0 = BasicBlock(ch_in, ch_out, data_format=data_format)
        self.res_blocks = nn.Sequential([BasicBlock(ch_out, ch_out, data_format=data_format) for _ in range(1, count)])
        self.ch_out = ch_out

    def run_nhwc(self, v):
        v = self.basicblock0.run_nhwc(v)
        for b in self.res_blocks:
            v = b.run_nhwc(v)
        return v

    def forward(self, inputs):
        return self.res_blocks(self.basicblock0(inputs))


DarkNet_cfg = {53: [1, 2, 8, 8, 4]}


class DarkNet(nn.Module):
    def __init__(self, depth=53, freeze_at=-1, return_idx=[2, 3, 4], num_stages=5, norm_type="bn", norm_decay=0.0,
                 freeze_norm=False, data_format="channels_first"):
        super().__init__()
        self.depth, self.freeze_at, self.return_idx, self.num_stages = depth, freeze_at, return_idx, num_stages
        self.stages = DarkNet_cfg[self.depth][0:num_stages]
        self.data_format = data_format
        self.conv0 = ConvBNLayer(ch_in=3, ch_out=32, filter_size=3, stride=1, padding=1, data_format=data_format)
        self.downsample0 = DownSample(ch_in=32, ch_out=32 * 2, data_format=data_format)
        self._out_channels = []
        # plain Python lists, as in the reference (darknet.py:270-297); Module adopts them on set_eval()
        self.darknet_conv_block_list = []
        self.downsample_list = []
        ch_in = [64, 128, 256, 512, 1024]
        for i, stage in enumerate(self.stages):
            self.darknet_conv_block_list.append(Blocks(int(ch_in[i]), int(ch_in[i]), stage, data_format=data_format,
                                                       name="stage.{}".format(i)))
            if i in return_idx:
                self._out_channels.append(int(ch_in[i]))
        for i in range(num_stages - 1):
            self.downsample_list.append(DownSample(ch_in=int(ch_in[i]), ch_out=int(ch_in[i + 1]), data_format=data_format))

    def run_nhwc(self, v):
        out = self.downsample0.run_nhwc(self.conv0.run_nhwc(v))
        blocks = []
        for i, blk in enumerate(self.darknet_conv_block_list):
            out = blk.run_nhwc(out)
            if i in self.return_idx:
                blocks.append(out)
            if i < self.num_stages - 1:
                out = self.downsample_list[i].run_nhwc(out)
        return blocks

    def forward(self, inputs):
        x = inputs["images"]                                             # darknet.py:300
        return [from_nhwc(b, self.data_format) for b in self.run_nhwc(as_nhwc(x, self.data_format))]
