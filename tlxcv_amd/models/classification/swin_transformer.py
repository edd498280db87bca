"""Swin Transformer (tiny/small/base/large, window 7) forward graph on the MI355X engine.

Same factories / constructor arguments / parameter tree as the reference
(tlxcv/models/classification/swin_transformer.py:119-683; that file is a Paddle conversion that
hard-imports `paddle`, so it is restated from its text).  Per block the reference does
LayerNorm -> reshape -> roll -> window_partition -> qkv Linear -> scaled q k^T + relative position
bias (+ shift mask) -> softmax -> @v -> proj -> window_reverse -> roll back -> residual -> LayerNorm
-> Mlp -> residual (:310-337, :192-229).  Here:
  * norm1 + roll + window_partition are one pass over the rows (tlxmi_layernorm_window_partition), and so are
    window_reverse + roll back + the residual add + norm2 (tlxmi_window_reverse_layernorm);
  * the attention core is one fused MFMA kernel on the packed qkv matrix that adds the pre-gathered
    (heads, 49, 49) bias table and the (nW, 49, 49) shift mask in registers;
  * qkv / proj / fc1(+GELU) / fc2(+residual) are the implicit-GEMM kernel with fused epilogues;
  * PatchMerging's strided 2x2 gather + concat + LayerNorm is one pass (tlxmi_patch_merge_layernorm), then the GEMM.
Round 5, fp16 at >= `lnfold_min_rows` token rows per launch and channel widths >= `lnfold_min_c` (stages 2 - 4 of
Swin-B at bench size): the residual stream stays in IMAGE order and neither LayerNorm-type pass runs — the producing GEMM (proj,
fc2, the PatchMerging reduction) leaves per-row (sum, sum of squares) per 256-channel tile column, qkv / fc1 run on the raw stream
with gamma folded into the weight, form mean / rstd of their rows from those and apply the per-row affine in the epilogue, and roll + window_partition / window_reverse + roll back are row
arithmetic inside the attention kernel (tlxmi_attention_windows).  Stage 1 (128 channels: statistics of a row fit no tile
economy, the passes stay) runs its Mlp as one launch with the hidden map on chip (tlxmi_mlp_seam).
Quirks kept on purpose: the -100.0 (not -inf) mask, and the PatchMerging reduction bias (:369-370).
"""

import numpy as np
import torch

from ... import engine as E
from ...tlx import nn
from ...tlx.nn import as_nhwc

__all__ = ["SwinTransformer", "window_partition", "window_reverse", "drop_path", "DropPath", "swintransformer_tiny_patch4_window7_224", "swintransformer_small_patch4_window7_224",
           "swintransformer_base_patch4_window7_224", "swintransformer_large_patch4_window7_224",
           "swintransformer_base_patch4_window12_384", "swintransformer_large_patch4_window12_384"]

trunc_normal_ = nn.initializers.TruncatedNormal(stddev=0.02)


def to_2tuple(x):
    return tuple([x] * 2)


def drop_path(x, drop_prob=0.0, training=False):
    """Stochastic depth (swin_transformer.py:35-47): the identity in eval mode, the only mode this engine runs."""
    if drop_prob == 0.0 or not training:
        return x
    raise RuntimeError("tlxcv_amd: drop_path in training mode — this engine runs eval-mode forward passes only")


class DropPath(nn.Module):
    """swin_transformer.py:50-59."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        return drop_path(x, self.drop_prob or 0.0, self.is_train)


def window_partition(x, window_size):
    """(B, H, W, C) -> (num_windows * B, window_size, window_size, C)   (swin_transformer.py:85-99), on tlxmi_window_partition."""
    B, H, W, C = x.shape
    x = E.need_gpu(x, "input").contiguous()
    return E.window_partition(x, int(window_size), 0).view(-1, window_size, window_size, C)


def window_reverse(windows, window_size, H, W, C):
    """(num_windows * B, window_size, window_size, C) -> (B, H, W, C)   (swin_transformer.py:102-116), on tlxmi_window_reverse."""
    windows = E.need_gpu(windows, "input").contiguous()
    B = windows.numel() // (H * W * C)
    return E.window_reverse(windows.view(-1, window_size * window_size, C), B, H, W, int(window_size), 0)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features=in_features, out_features=hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(in_features=hidden_features, out_features=out_features)
        self.drop = nn.Dropout(drop)

    def run(self, x, res=None):
        dt = E.precision()
        if (res is not None and getattr(self.act, "ACT", None) == E.ACT_GELU and x.dtype == dt
                and E.mlp_seam_supported(x.numel() // x.shape[-1], self.fc1.in_features, self.fc1.out_features, self.fc2.out_features, dt)):
            # fc1 + GELU + fc2 + residual as one launch: the (rows, 4 C) hidden map stays on the CU (round 5; stage 1 of Swin-B)
            pk1 = self.fc1._cached("pk", lambda: E.PackedFilter(self.fc1.weights.detach().t().contiguous(), dt))
            pk2 = self.fc2._cached("pk", lambda: E.PackedFilter(self.fc2.weights.detach().t().contiguous(), dt))
            b1 = self.fc1._cached("bias", lambda: E._f32(self.fc1.biases)) if self.fc1.biases is not None else None
            b2 = self.fc2._cached("bias", lambda: E._f32(self.fc2.biases)) if self.fc2.biases is not None else None
            return E.mlp_seam(x, pk1, b1, pk2, b2, res, out=res)
        return self.fc2.run(self.fc1.run(x, act=self.act.ACT), res=res, out=res)

    def forward(self, x):
        return self.run(x)


def relative_position_index(ws):
    """swin_transformer.py:146-158."""
    coords = torch.stack(torch.meshgrid(torch.arange(ws[0]), torch.arange(ws[1]), indexing="ij"))
    cf = torch.flatten(coords, 1)
    rel = (cf.unsqueeze(2) - cf.unsqueeze(1)).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws[0] - 1
    rel[:, :, 1] += ws[1] - 1
    rel[:, :, 0] *= 2 * ws[1] - 1
    return rel.sum(-1)


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.relative_position_bias_table = nn.Parameter(
            data=trunc_normal_(shape=((2 * window_size[0] - 1) * (2 * window_size[1] - 1), num_heads)))
        self.register_buffer("relative_position_index", relative_position_index(window_size))
        self.qkv = nn.Linear(in_features=dim, out_features=dim * 3, b_init="constant" if qkv_bias else None)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(in_features=dim, out_features=dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def bias_table(self):
        """(heads, N, N) fp32, gathered once (the reference caches the same tensor in eval(), :179-190)."""
        def build():
            n = self.window_size[0] * self.window_size[1]
            idx = self.relative_position_index.reshape(-1)
            b = torch.index_select(self.relative_position_bias_table.detach(), 0, idx)
            return b.reshape(n, n, -1).permute(2, 0, 1).contiguous().float()
        return self._cached("rpb", build)

    def run(self, xw, mask=None):
        qkv = self.qkv.run(xw)                                                     # (B_, N, 3C)
        hd = self.dim // self.num_heads
        if qkv.dtype == torch.float16 and hd in (32, 64, 96) and qkv.shape[1] <= 256 and E.option("attn_comb"):
            # relative position bias + shift mask summed and padded once per layer (:205-220 adds them per forward)
            # keyed by the mask's storage, in-place version and shape — not id(): a temporary mask can be collected and its
            # id reused by another tensor
            mkey = None if mask is None else (mask.data_ptr(), mask._version, tuple(mask.shape))
            tab = self._cached(("rpb+mask", mkey), lambda: E.attention_table(self.bias_table(), mask, qkv.shape[1]))
            a = E.attention_comb(qkv, self.num_heads, self.scale, tab, 0 if mask is None else mask.shape[0])
        else:
            a = E.attention(qkv, self.num_heads, self.scale, self.bias_table(), mask)  # :202-226
        return self.proj.run(a)

    def table(self, mask, n):
        """bias (+ shift mask) summed and padded once per layer (tlxmi_attention_comb's table; :205-220 adds the two per forward)."""
        mkey = None if mask is None else (mask.data_ptr(), mask._version, tuple(mask.shape))
        return self._cached(("rpb+mask", mkey), lambda: E.attention_table(self.bias_table(), mask, n))

    def forward(self, x, mask=None):
        return self.run(x.to(E.precision()).contiguous(), mask)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0, mlp_ratio=4.0, qkv_bias=True,
                 qk_scale=None, drop=0.0, attn_drop=0.0, drop_path=0.0, act_layer=nn.GELU, layer_norm=nn.LayerNorm):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        self.window_size, self.shift_size, self.mlp_ratio = window_size, shift_size, mlp_ratio
        if min(self.input_resolution) <= self.window_size:                         # :274-276
            self.shift_size = 0
            self.window_size = min(self.input_resolution)
        assert 0 <= self.shift_size < self.window_size, 'shift_size must in 0-window_size'
        self.norm1 = layer_norm(dim)
        self.attn = WindowAttention(dim, window_size=to_2tuple(self.window_size), num_heads=num_heads,
                                    qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = nn.Identity()
        self.norm2 = layer_norm(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        if self.shift_size > 0:                                                    # :288-305
            H, W = self.input_resolution
            ws, ss = self.window_size, self.shift_size
            img_mask = torch.zeros((1, H, W, 1))
            cnt = 0
            for h in (slice(0, -ws), slice(-ws, -ss), slice(-ss, None)):
                for w in (slice(0, -ws), slice(-ws, -ss), slice(-ss, None)):
                    img_mask[:, h, w, :] = cnt
                    cnt += 1
            mw = img_mask.reshape(1, H // ws, ws, W // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws)
            am = mw.unsqueeze(1) - mw.unsqueeze(2)
            attn_mask = -100.0 * (am != 0).float()
        else:
            attn_mask = None
        self.register_buffer("attn_mask", attn_mask)

    def run(self, x):
        """x (B, L, C) engine dtype -> new (B, L, C)."""
        H, W = self.input_resolution
        B, L, C = x.shape
        assert L == H * W, 'input feature has wrong size'
        # norm1 + roll + window_partition in one pass (:315-324)
        win = E.layernorm_window_partition(x.view(B, H, W, C), self.norm1.gamma.detach(), self.norm1.beta.detach(),
                                           self.norm1.epsilon, self.window_size, self.shift_size)
        aw = self.attn.run(win, self.attn_mask)                                    # :325
        # window_reverse + roll back + residual + norm2 in one pass (:327-335)
        x, h2 = E.window_reverse_layernorm(aw, x.view(B, H, W, C), self.norm2.gamma.detach(), self.norm2.beta.detach(),
                                           self.norm2.epsilon, self.window_size, self.shift_size)
        x = x.view(B, L, C)
        self.mlp.run(h2.view(B, L, C), res=x)                                      # :335, in place
        return x

    # ---- round 5: the block without its two LayerNorm-type passes.  The residual stream stays in IMAGE order all the way: the
    # LayerNorms are folded around the Linear layers (row statistics out of the proj / fc2 epilogues, the normalisation in the qkv / fc1
    # epilogues: engine.linear_stats / linear_ln), and roll + window_partition / window_reverse + roll back (:316-333) are the row
    # arithmetic of the attention kernel (engine.attention_windows) instead of two passes over the activations.
    def folded_ok(self, x):
        return self.folded_ok_shape(x.shape[0] * x.shape[1], x.shape[2], x.dtype)

    def folded_ok_shape(self, rows, C, dtype):
        a, m, ws = self.attn, self.mlp, self.window_size
        hd = C // self.num_heads
        if not (E.option("lnfold") and E.option("attn_comb") and dtype == torch.float16 and hd in (32, 64, 96) and ws * ws <= 64
                and isinstance(self.norm1, nn.LayerNorm) and isinstance(self.norm2, nn.LayerNorm) and hasattr(m.act, "ACT")
                and m.act.ACT in (E.ACT_NONE, E.ACT_GELU) and C >= E.option_value("lnfold_min_c")):
            return False
        hid = m.fc1.out_features
        return (E.linear_ln_supported(rows, C, 3 * C, dtype) and E.linear_ln_supported(rows, C, C, dtype, with_res=True)
                and E.linear_ln_supported(rows, C, hid, dtype, act=m.act.ACT) and E.linear_ln_supported(rows, hid, C, dtype, with_res=True))

    def run_folded(self, x, part, stats=True):
        """x (B, L, C) fp16 updated IN PLACE; part: partial row statistics of x (engine.linear_stats of whoever wrote x); returns those of
        the new x (None when stats is False)."""
        H, W = self.input_resolution
        B, L, C = x.shape
        assert L == H * W, 'input feature has wrong size'
        a, ws = self.attn, self.window_size
        qkv = a.qkv.run_ln(x, self.norm1, part)                       # norm1 + qkv (:315, :194)
        tab = a.table(self.attn_mask, ws * ws)
        aw = E.attention_windows(qkv, a.num_heads, a.scale, tab, 0 if self.attn_mask is None else self.attn_mask.shape[0], H, W, ws,
                                 self.shift_size)                                                            # :316-333 around :202-226
        part = a.proj.run_stats(aw, res=x, out=x)[1]                                                         # proj + shortcut (:226, :334)
        h = self.mlp.fc1.run_ln(x, self.norm2, part, act=self.mlp.act.ACT)   # norm2 + fc1 + GELU
        if stats:
            return self.mlp.fc2.run_stats(h, res=x, out=x)[1]                                                # fc2 + residual (:335)
        self.mlp.fc2.run(h, res=x, out=x)
        return None

    def forward(self, x):
        E.need_gpu(x, "input")
        return self.run(x.to(E.precision()).contiguous())


class PatchMerging(nn.Module):
    def __init__(self, input_resolution, dim, layer_norm=nn.LayerNorm):
        super().__init__()
        self.input_resolution, self.dim = input_resolution, dim
        self.reduction = nn.Linear(in_features=4 * dim, out_features=2 * dim, b_init=nn.initializers.xavier_uniform())
        self.norm = layer_norm(4 * dim)

    def run(self, x, stats=False):
        """stats: also return the partial row statistics of the output (engine.linear_stats) when the reduction takes that path:
        (y, part) instead of y."""
        H, W = self.input_resolution
        B, L, C = x.shape
        assert L == H * W, 'input feature has wrong size'
        assert H % 2 == 0 and W % 2 == 0, 'x size ({}*{}) are not even.'.format(H, W)
        if isinstance(self.norm, nn.LayerNorm):      # gather + concat + norm in one pass (:381-388)
            g = E.patch_merge_layernorm(x.view(B, H, W, C), self.norm.gamma.detach(), self.norm.beta.detach(), self.norm.epsilon)
            if stats and E.linear_ln_supported(B * L // 4, 4 * C, self.reduction.out_features, g.dtype, producer=True):
                return self.reduction.run_stats(g)                                 # + the row statistics for the next stage's first norm1
            return self.reduction.run(g)                                           # :389
        g = E.patch_merge_gather(x.view(B, H, W, C)).view(B, H * W // 4, 4 * C)    # :381-387
        return self.reduction.run(self.norm(g))                                    # :388-389

    def forward(self, x):
        return self.run(x.to(E.precision()).contiguous())


class BasicLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4.0, qkv_bias=True,
                 qk_scale=None, drop=0.0, attn_drop=0.0, drop_path=0.0, layer_norm=nn.LayerNorm, downsample=None,
                 use_checkpoint=False):
        super().__init__()
        self.dim, self.input_resolution, self.depth = dim, input_resolution, depth
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim=dim, input_resolution=input_resolution, num_heads=num_heads,
                                 window_size=window_size, shift_size=0 if i % 2 == 0 else window_size // 2,
                                 mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop,
                                 attn_drop=attn_drop, layer_norm=layer_norm) for i in range(depth)])
        self.downsample = downsample(input_resolution, dim=dim, layer_norm=layer_norm) if downsample is not None else None

    def run(self, x, part=None, next_folded=False):
        """part: partial row statistics of x when its producer emitted them (the previous stage's PatchMerging) -> the blocks run without
        LayerNorm / window passes where they can (SwinTransformerBlock.run_folded).  next_folded: ask the PatchMerging for the statistics
        of its output.  Returns (x, part of the output or None)."""
        nb = len(self.blocks)
        for i, blk in enumerate(self.blocks):
            if part is not None and blk.folded_ok(x):
                last = i == nb - 1 or not self.blocks[i + 1].folded_ok(x)
                part = blk.run_folded(x, part, stats=not last)
            else:
                x = blk.run(x)
                part = None
        part = None
        if self.downsample is not None:
            y = self.downsample.run(x, stats=next_folded)
            x, part = y if isinstance(y, tuple) else (y, None)
        return x, part

    def forward(self, x):
        return self.run(x.to(E.precision()).contiguous())[0]


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, layer_norm=None):
        super().__init__()
        img_size, patch_size = to_2tuple(img_size), to_2tuple(patch_size)
        self.patches_resolution = [img_size[0] // patch_size[0], img_size[1] // patch_size[1]]
        self.img_size, self.patch_size = img_size, patch_size
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.proj = nn.GroupConv2d(kernel_size=patch_size, stride=patch_size, in_channels=in_chans,
                                   out_channels=embed_dim, padding=0, data_format='channels_first')
        self.norm = layer_norm(embed_dim) if layer_norm is not None else None

    def forward(self, x, pos=None):
        """pos: fp32 (num_patches, D) absolute position embedding added to the tokens after the norm (SwinTransformer(ape=True), :603-604)."""
        conv, D = self.proj, self.embed_dim
        if (E.option("patch_embed4") and E.precision() == torch.float16 and tuple(self.patch_size) == (4, 4) and self.in_chans == 3
                and D in (96, 128, 192, 256) and x.dim() == 4 and x.shape[1] == 3 and x.shape[2] % 4 == 0 and x.shape[3] % 4 == 0 and x.dtype in (torch.float16, torch.float32)
                and not x.permute(0, 2, 3, 1).is_contiguous() and (self.norm is None or isinstance(self.norm, nn.LayerNorm))):
            # conv + flatten + transpose + norm (:498-504) in one pass over the image: the layer is HBM traffic only
            w64 = conv._cached("pe4", lambda: E.patch_embed4_filter(conv.filters))
            bias = conv._cached("bias", lambda: E._f32(conv.biases)) if conv.biases is not None else None
            if self.norm is None:
                return E.patch_embed4(x, w64, bias, None, None, 0.0, pos)
            return E.patch_embed4(x, w64, bias, self.norm.gamma.detach(), self.norm.beta.detach(), self.norm.epsilon, pos)
        if self.patch_size[0] % 4 == 0 and not x.permute(0, 2, 3, 1).is_contiguous():
            y = self.proj.run_stem(x, 4)                                           # 4x4/4 conv == 1x1 conv on 48 folded channels
        else:
            y = self.proj.run_nhwc(as_nhwc(x, 'channels_first'))                   # (B, H/4, W/4, D): :500-501
        y = y.view(y.shape[0], -1, y.shape[-1])
        y = self.norm(y) if self.norm is not None else y
        if pos is not None:      # x + absolute_pos_embed, one table for every image: a per-element shift over the flattened image
            B, L, _ = y.shape
            y = E.affine_act(y.contiguous().view(B, L * D), shift=pos.reshape(-1)).view(B, L, D)
        return y


class SwinTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=4, in_chans=3, class_num=1000, embed_dim=96, depths=[2, 2, 6, 2],
                 num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop_rate=0.0,
                 attn_drop_rate=0.0, drop_path_rate=0.1, layer_norm=nn.LayerNorm, ape=False, patch_norm=True,
                 use_checkpoint=False, **kwargs):
        super().__init__()
        self.num_classes = num_classes = class_num
        self.num_layers = len(depths)
        self.embed_dim, self.ape, self.patch_norm = embed_dim, ape, patch_norm
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.mlp_ratio = mlp_ratio
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                      layer_norm=layer_norm if self.patch_norm else None)
        pr = self.patches_resolution = self.patch_embed.patches_resolution
        if self.ape:      # swin_transformer.py:561-565 (no shipped config sets it)
            self.absolute_pos_embed = nn.Parameter(data=trunc_normal_(shape=(1, self.patch_embed.num_patches, embed_dim)))
            self.register_parameter(name="absolute_pos_embed", param=self.absolute_pos_embed)
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(BasicLayer(
                dim=int(embed_dim * 2 ** i), input_resolution=(pr[0] // 2 ** i, pr[1] // 2 ** i), depth=depths[i],
                num_heads=num_heads[i], window_size=window_size, mlp_ratio=self.mlp_ratio, qkv_bias=qkv_bias,
                qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate, layer_norm=layer_norm,
                downsample=PatchMerging if i < self.num_layers - 1 else None))
        self.norm = layer_norm(self.num_features)
        self.avgpool = nn.AdaptiveAvgPool1d(1, data_format='channels_first')
        self.head = nn.Linear(in_features=self.num_features, out_features=num_classes) if num_classes > 0 else nn.Identity()

    def forward_features(self, x):
        self._require_eval()
        pos = None
        if self.ape:                                   # :603-604, folded into the patch embedding's store
            pos = self._cached("ape", lambda: E._f32(self.absolute_pos_embed.detach()[0]).contiguous())
        x = self.patch_embed(x, pos)                   # :602
        part = None
        for i, layer in enumerate(self.layers):        # :606-607
            nxt = self.layers[i + 1] if i + 1 < len(self.layers) else None
            # (the next stage's first block decides from the shape its input will have)
            want = nxt is not None and nxt.blocks[0].folded_ok_shape(x.shape[0] * x.shape[1] // 4, x.shape[2] * 2, x.dtype)
            x, part = layer.run(x, part, next_folded=want)
        x = self.norm(x)                               # :608
        return E.global_avgpool(x)                     # mean over tokens == avgpool(x^T) + flatten, :609-610

    # tools/two_stream_threshold.py (round 5, profiles/r05/two_stream_threshold.txt; one stream / halves with no plan flag / planned for the
    # device): batch 16 2.41 / 2.36 / 3.12 ms, 32 3.06 / 2.89 / 3.04, 64 4.18 / 3.87 / 3.99, 96 5.48 / 5.27 / 5.32, 128 6.80 / 6.53 / 6.56
    @E.two_streams(32, plan=None)
    def forward(self, x):
        x = self.forward_features(x)
        return self.head.run(x) if isinstance(self.head, nn.Linear) else x


_CFG = {
    'swintransformer_tiny_patch4_window7_224': dict(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7, drop_path_rate=0.2),
    'swintransformer_small_patch4_window7_224': dict(embed_dim=96, depths=[2, 2, 18, 2], num_heads=[3, 6, 12, 24], window_size=7),
    'swintransformer_base_patch4_window7_224': dict(embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=7, drop_path_rate=0.5),
    'swintransformer_large_patch4_window7_224': dict(embed_dim=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48], window_size=7),
    # 384 x 384 input, 12 x 12 windows = 144 tokens per window (swin_transformer.py:641-650); stage 4 is one window (no shift)
    'swintransformer_base_patch4_window12_384': dict(img_size=384, embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32],
                                                     window_size=12, drop_path_rate=0.5),
    'swintransformer_large_patch4_window12_384': dict(img_size=384, embed_dim=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48],
                                                      window_size=12),
}


def _swin(arch, pretrained=False, use_ssld=False, **kwargs):
    if pretrained:
        raise NotImplementedError("pretrained weights are not bundled; use model.load_weights(...)")
    return SwinTransformer(**{**_CFG[arch], **kwargs})


def swintransformer_tiny_patch4_window7_224(pretrained=False, use_ssld=False, **kwargs):
    return _swin('swintransformer_tiny_patch4_window7_224', pretrained, use_ssld, **kwargs)


def swintransformer_small_patch4_window7_224(pretrained=False, use_ssld=False, **kwargs):
    return _swin('swintransformer_small_patch4_window7_224', pretrained, use_ssld, **kwargs)


def swintransformer_base_patch4_window7_224(pretrained=False, use_ssld=False, **kwargs):
    return _swin('swintransformer_base_patch4_window7_224', pretrained, use_ssld, **kwargs)


def swintransformer_large_patch4_window7_224(pretrained=False, use_ssld=False, **kwargs):
    return _swin('swintransformer_large_patch4_window7_224', pretrained, use_ssld, **kwargs)


def swintransformer_base_patch4_window12_384(pretrained=False, use_ssld=False, **kwargs):
    return _swin('swintransformer_base_patch4_window12_384', pretrained, use_ssld, **kwargs)


def swintransformer_large_patch4_window12_384(pretrained=False, use_ssld=False, **kwargs):
    return _swin('swintransformer_large_patch4_window12_384', pretrained, use_ssld, **kwargs)
