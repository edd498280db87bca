"""Vision Transformer forward graph on the MI355X engine.

Same factories / constructor arguments / parameter tree as the reference
(tlxcv/models/classification/vision_transformer.py:64-447).  Forward differences (all fusions, no
change of arithmetic):
  * PatchEmbed conv + flatten + transpose + concat cls + add pos_embed (:206-220, :321-323) is ONE Linear over all B * (1 + P)
    token rows: patch rows in the flattened filter's order with a zero row in each image's cls slot (tlxmi_patchify), the conv
    filter as the weight, the per-image residual rows [cls + pos[0] | pos[1:] + bias] broadcast into the token matrix first and added
    in place (round 4; round 5: one (1 + P, D) table per model instead of a (B, 1 + P, D) tensor per batch size; patch sizes that
    are multiples of 8).
    Otherwise (and with set_option("patch_linear", False)): the conv on a space-to-depth image writes straight into rows 1..P of
    the token matrix with `+ pos_embed[1:]` in its epilogue, and row 0 is the constant cls_token + pos_embed[0].
  * Attention (:112-123) = qkv GEMM(+bias) -> one fused softmax(q k^T * scale) v kernel on the
    packed qkv matrix -> proj GEMM with bias and the residual add of Block.forward (:173) fused.
  * Mlp (:81-87) = fc1 GEMM with bias+exact-erf GELU epilogue -> fc2 GEMM with bias + residual (:174).
  * norm1 / norm2 (:172-174) are no launches and no passes over the token matrix at fp16 bench sizes (round 5): the
    GEMM that writes the residual stream (patch Linear, proj, fc2) leaves per-row (sum, sum of squares) partials from its fp32
    epilogue values (one pair per 256-channel tile column), and qkv / fc1 run on the RAW stream with gamma folded into the weight,
    form (a, b) = (rstd, -mean * rstd) per row from those pairs and store  y = a * acc + b * c1[n] + c2[n].  Below `lnfold_min_rows` token rows per launch,
    in fp32, and with set_option("lnfold", 0): LayerNorm launches of their own in front of the qkv / fc1 GEMMs.
  * final LayerNorm only on the cls rows (x[:, 0] commutes with a per-row norm, :327-328).
"""
import numpy as np
import torch

from ... import engine as E
from ...tlx import nn
from ...tlx.nn import as_nhwc

__all__ = ["VisionTransformer", "vit_small_patch16_224", "vit_base_patch16_224", "vit_base_patch16_384",
           "vit_base_patch32_384", "vit_large_patch16_224", "vit_large_patch16_384", "vit_large_patch32_384"]

trunc_normal_ = nn.initializers.TruncatedNormal(stddev=0.02)
zeros_ = nn.initializers.Constant(value=0.0)
ones_ = nn.initializers.Constant(value=1.0)


def to_2tuple(x):
    return (x, x)


class Identity(nn.Module):
    def forward(self, x):
        return x


class DropPath(nn.Module):
    """Stochastic depth: identity in eval (vision_transformer.py:36-61)."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob and self.is_train:
            self._require_eval()
        return x


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features=in_features, out_features=hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(in_features=hidden_features, out_features=out_features)
        self.drop = nn.Dropout(drop)

    def run(self, x, res=None, norm=None):
        # `norm` (Block.norm2), then bias + GELU in the GEMM epilogue
        h = self.fc1.run(norm(x) if norm is not None else x, act=self.act.ACT)
        return self.fc2.run(h, res=res, out=res)       # bias + residual, written in place

    def run_folded(self, x, norm, part, stats=True):
        """x += fc2(act(fc1(norm(x)))) with `norm` applied in fc1's epilogue from the row statistics `part` of x; returns the partial
        row statistics of the new x out of fc2's epilogue (None when stats is False: the last block)."""
        h = self.fc1.run_ln(x, norm, part, act=self.act.ACT)
        if stats:
            return self.fc2.run_stats(h, res=x, out=x)[1]
        self.fc2.run(h, res=x, out=x)
        return None

    def forward(self, x):
        return self.run(x)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        if qkv_bias:
            self.qkv = nn.Linear(in_features=dim, out_features=dim * 3)
        else:
            self.qkv = nn.Linear(in_features=dim, out_features=dim * 3, b_init=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(in_features=dim, out_features=dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def run(self, x, res=None, norm=None):
        # `norm` (Block.norm1), then (B, N, 3*C), packed [3][heads][hd]
        qkv = self.qkv.run(norm(x) if norm is not None else x)
        a = E.attention(qkv, self.num_heads, self.scale)               # softmax(q k^T * scale) v, heads merged
        return self.proj.run(a, res=res, out=res)

    def run_folded(self, x, norm, part):
        """x += proj(attention(qkv(norm(x)))) with `norm` applied in qkv's epilogue from the row statistics `part` of x; returns the
        partial row statistics of the new x out of proj's epilogue."""
        qkv = self.qkv.run_ln(x, norm, part)
        a = E.attention(qkv, self.num_heads, self.scale)
        return self.proj.run_stats(a, res=x, out=x)[1]

    def forward(self, x):
        return self.run(x)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, act_layer=nn.GELU, layer_norm="nn.LayerNorm", epsilon=1e-05,
                 data_format="channels_first"):
        super().__init__()
        if isinstance(layer_norm, str):
            self.norm1 = eval(layer_norm)(dim, epsilon=epsilon)
        elif callable(layer_norm):
            self.norm1 = layer_norm(dim)
        else:
            raise TypeError("The layer_norm must be str or paddle.nn.layer.Layer class")
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else Identity()
        if isinstance(layer_norm, str):
            self.norm2 = eval(layer_norm)(dim, epsilon=epsilon)
        else:
            self.norm2 = layer_norm(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def run_inplace(self, x):
        """x (B, N, C) engine dtype, updated in place: x += attn(norm1(x)); x += mlp(norm2(x))."""
        self.attn.run(x, res=x, norm=self.norm1)
        self.mlp.run(x, res=x, norm=self.norm2)
        return x

    def folded_ok(self, x):
        """Round 5: the two LayerNorm launches of a block replaced by row statistics out of the producing GEMM's epilogue and a row
        affine in the consuming GEMM's (engine.linear_stats / linear_ln; vision_transformer.py:172-175)."""
        rows, D = x.shape[0] * x.shape[1], x.shape[2]
        a, m = self.attn, self.mlp
        if not (isinstance(self.norm1, nn.LayerNorm) and isinstance(self.norm2, nn.LayerNorm) and hasattr(m.act, "ACT")):
            return False
        hid = m.fc1.out_features
        return (E.linear_ln_supported(rows, D, 3 * D, x.dtype) and E.linear_ln_supported(rows, D, D, x.dtype, with_res=True)
                and E.linear_ln_supported(rows, D, hid, x.dtype, act=m.act.ACT) and E.linear_ln_supported(rows, hid, D, x.dtype, with_res=True)
                and a.qkv.out_features == 3 * D and m.act.ACT in (E.ACT_NONE, E.ACT_GELU))

    def run_folded(self, x, part, last=False):
        """run_inplace with the LayerNorms folded: `part` = partial row statistics of x (from whoever wrote x); returns those of the
        new x (None for the last block)."""
        D = x.shape[2]
        part = self.attn.run_folded(x, self.norm1, part)
        return self.mlp.run_folded(x, self.norm2, part, stats=not last)

    def forward(self, x):
        E.need_gpu(x, "input")
        x = x.to(E.precision()).contiguous().clone()
        return self.run_inplace(x)


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, data_format="channels_first"):
        super().__init__()
        img_size = to_2tuple(img_size)
        patch_size = to_2tuple(patch_size)
        self.num_patches = img_size[1] // patch_size[1] * (img_size[0] // patch_size[0])
        self.img_size, self.patch_size, self.data_format = img_size, patch_size, data_format
        self.proj = nn.GroupConv2d(kernel_size=patch_size, stride=patch_size, in_channels=in_chans,
                                   out_channels=embed_dim, padding=0, data_format=data_format)

    def check(self, x):
        H, W = (x.shape[2], x.shape[3]) if self.data_format == "channels_first" else (x.shape[1], x.shape[2])
        assert H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."

    def forward(self, x):
        self.check(x)
        y = self.proj.run_nhwc(as_nhwc(x, self.data_format))
        return y.reshape(y.shape[0], -1, y.shape[-1])


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4, qkv_bias=False, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0,
                 drop_path_rate=0.0, layer_norm="nn.LayerNorm", epsilon=1e-05, data_format="channels_first",
                 name=None):
        super().__init__(name=name)
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.data_format = data_format
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                      embed_dim=embed_dim, data_format=data_format)
        num_patches = self.patch_embed.num_patches
        self.pos_embed = self.add_parameter("pos_embed", shape=(1, num_patches + 1, embed_dim),
                                            initializer=trunc_normal_)
        self.cls_token = self.add_parameter("cls_token", shape=(1, 1, embed_dim), initializer=trunc_normal_)
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr, layer_norm=layer_norm, epsilon=epsilon)
            for dpr in np.linspace(0, drop_path_rate, depth)])
        self.norm = eval(layer_norm)(embed_dim, epsilon=epsilon)
        self.head = nn.Linear(in_features=embed_dim, out_features=num_classes) if num_classes > 0 else Identity()

    def add_parameter(self, name, shape, initializer=None):
        initializer = initializer or zeros_
        param = nn.Parameter(data=initializer(shape=shape))
        self.register_parameter(name=name, param=param)
        return param

    def forward_features(self, x):
        self._require_eval()
        pe = self.patch_embed
        pe.check(x)
        dt = E.precision()
        B, P, D = x.shape[0], pe.num_patches, self.embed_dim
        tok = torch.empty((B, P + 1, D), dtype=dt, device=x.device)
        ps, conv = pe.patch_size[0], pe.proj
        if (E.option("patch_linear") and self.data_format == "channels_first" and pe.patch_size[0] == pe.patch_size[1] and ps % 8 == 0 and B * (P + 1) < (1 << 20)
                and not x.permute(0, 2, 3, 1).is_contiguous() and (x.shape[1] * ps * ps) % 8 == 0
                and tuple(conv.stride) == tuple(pe.patch_size) and tuple(conv.padding) == (0, 0) and not conv.same):
            # The patch-embedding conv as ONE Linear over all B * (1 + P) token rows (vision_transformer.py:197-204, 321-323): patch
            # rows in the flattened filter's order with a zero row in the cls slot (tlxmi_patchify), the filter as the Linear weight,
            # and the rows [cls + pos[0] | pos[1:] + bias] as an in-place residual — the zero row yields cls + pos[0].  It runs on the
            # persistent GEMM like proj (same shape) instead of the generic implicit GEMM: ViT-B/16 batch 256 forward 11.01 -> 10.93 ms.
            pk = conv._cached(("patch_linear", dt), lambda: E.PackedFilter(conv.filters.detach().reshape(D, -1).contiguous(), dt))

            def rows_table():
                # one image's residual rows [cls + pos[0] | pos[1:] + bias], summed in fp32 and rounded ONCE (the conv bias rides here, not
                # in the GEMM epilogue: a zero patch row + this row is exactly cls + pos[0], whatever the bias is)
                first = self.cls_token.detach()[0, 0].float() + self.pos_embed.detach()[0, 0].float()
                rest = self.pos_embed.detach()[0, 1:].float()
                if conv.biases is not None:
                    rest = rest + conv.biases.detach().float()[None]
                return torch.cat((first[None], rest), 0).to(dt).contiguous().view(-1)             # ((1 + P) * D,)
            # (1 + P, D) per model, whatever the batch: the table is broadcast into the token matrix (one 2-byte-per-element write) and
            # the GEMM adds it as its in-place residual.  (Round 4 cached the expanded (B, 1 + P, D) tensor per batch size seen.)
            table = self._cached(("patch_rows", dt), rows_table, deps=(conv,))
            E.broadcast_rows_into(table, tok, B, (P + 1) * D)
            if len(self.blocks) and all(blk.folded_ok(tok) for blk in self.blocks) and E.linear_ln_supported(B * (P + 1), pk.Cin, D, dt, with_res=True):
                # LayerNorm statistics ride in the epilogues of the GEMMs that write the token matrix (round 5): no LayerNorm launch
                part = E.linear_stats(E.patchify(x, ps, 1, dt), pk, None, res=tok, out=tok)[1]
                for i, blk in enumerate(self.blocks):
                    part = blk.run_folded(tok, part, last=i == len(self.blocks) - 1)
                return E.layernorm_rows(tok, B, D, (P + 1) * D, self.norm.gamma.detach(), self.norm.beta.detach(), self.norm.epsilon)
            E.linear(E.patchify(x, ps, 1, dt), pk, None, res=tok, out=tok)
            for blk in self.blocks:
                blk.run_inplace(tok)
            return E.layernorm_rows(tok, B, D, (P + 1) * D, self.norm.gamma.detach(), self.norm.beta.detach(), self.norm.epsilon)
        # rows 1..P: conv + bias + pos_embed[1:]   (vision_transformer.py:206-220, 321-323)
        pos = self._cached("pos", lambda: self.pos_embed.detach()[0, 1:].to(dt).contiguous())            # (P, D)
        row0 = self._cached("row0", lambda: (self.cls_token.detach()[0, 0] + self.pos_embed.detach()[0, 0])
                            .to(dt).contiguous())                                                         # (D,)
        kw = dict(res=pos, out=tok[:, 1:], out_ld=D, y_nstride=(P + 1) * D, res_bcast=True, res_ld=D)
        fold = 4 if pe.patch_size[0] % 4 == 0 else (2 if pe.patch_size[0] % 2 == 0 else 0)
        if self.data_format == "channels_first" and fold and not x.permute(0, 2, 3, 1).is_contiguous():
            pe.proj.run_stem(x, fold, **kw)                    # K = 3*16*16 = 768 dense instead of 8*16*16
        else:
            pe.proj.run_nhwc(as_nhwc(x, self.data_format), **kw)
        # row 0: cls_token + pos_embed[0]
        E.broadcast_rows_into(row0, tok, B, (P + 1) * D)
        for blk in self.blocks:
            blk.run_inplace(tok)
        cls = E.layernorm_rows(tok, B, D, (P + 1) * D, self.norm.gamma.detach(), self.norm.beta.detach(),
                               self.norm.epsilon)                                                         # :327-328
        return cls

    # two half batches on two streams (round 3, hipGraph replay of ViT-B/16 on one box): batch 256 11.05 -> 10.76 ms with
    # the tiles planned for half the CUs, batch 128 6.14 -> 5.56 ms and batch 64 3.45 -> 3.35 ms with the device's own plan
    # round 5 (tools/plan_modes.py, with the LayerNorms folded into the GEMMs): batch 256 one stream 10.18 ms, halves planned for the device 9.85, for
    # half the CUs 10.14 (round 3 had it the other way round: the stand-alone LayerNorm launches filled the gaps the half plan left); batch 128: 5.69 / 5.17 / 5.77
    # end of round 5 (tools/two_stream_threshold.py, profiles/r05/two_stream_threshold.txt; one stream / halves with no plan flag / planned
    # for the device): batch 16 1.35 / 1.37 / 1.83 ms, 32 2.07 / 1.90 / 1.92, 64 2.90 / 2.88 / 2.86, 128 5.36 / 4.97 / 5.00, 256 9.75 / 9.70 / 9.69
    # -> halves from 32 images, each launch planned as if alone
    @E.two_streams(32, plan=None)
    def forward(self, x):
        x = self.forward_features(x)
        if isinstance(self.head, nn.Linear):
            return self.head.run(x)
        return x


_CFG = {
    "vit_small_patch16_224": dict(patch_size=16, embed_dim=768, depth=8, num_heads=8, mlp_ratio=3, qk_scale=768 ** -0.5),
    "vit_base_patch16_224": dict(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True, epsilon=1e-06),
    "vit_base_patch16_384": dict(img_size=384, patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True, epsilon=1e-06),
    "vit_base_patch32_384": dict(img_size=384, patch_size=32, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True, epsilon=1e-06),
    "vit_large_patch16_224": dict(patch_size=16, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4, qkv_bias=True, epsilon=1e-06),
    "vit_large_patch16_384": dict(img_size=384, patch_size=16, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4, qkv_bias=True, epsilon=1e-06),
    "vit_large_patch32_384": dict(img_size=384, patch_size=32, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4, qkv_bias=True, epsilon=1e-06),
}


def _vision_transformer(arch, pretrained, **kwargs):
    if pretrained:
        print("Warn: NotImplemented")
    return VisionTransformer(**{**_CFG[arch], **kwargs})


def vit_small_patch16_224(pretrained=False, **kwargs):
    return _vision_transformer("vit_small_patch16_224", pretrained, **kwargs)


def vit_base_patch16_224(pretrained=False, use_ssld=False, **kwargs):
    return _vision_transformer("vit_base_patch16_224", pretrained, **kwargs)


def vit_base_patch16_384(pretrained=False, use_ssld=False, **kwargs):
    return _vision_transformer("vit_base_patch16_384", pretrained, **kwargs)


def vit_base_patch32_384(pretrained=False, use_ssld=False, **kwargs):
    return _vision_transformer("vit_base_patch32_384", pretrained, **kwargs)


def vit_large_patch16_224(pretrained=False, use_ssld=False, **kwargs):
    return _vision_transformer("vit_large_patch16_224", pretrained, **kwargs)


def vit_large_patch16_384(pretrained=False, use_ssld=False, **kwargs):
    return _vision_transformer("vit_large_patch16_384", pretrained, **kwargs)


def vit_large_patch32_384(pretrained=False, use_ssld=False, **kwargs):
    return _vision_transformer("vit_large_patch32_384", pretrained, **kwargs)
