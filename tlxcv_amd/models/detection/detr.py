"""DETR's MultiHeadAttention on the MI355X engine — same class, constructor and parameter tree as
tlxcv/models/detection/detr.py:965-1062 (the second consumer of the fused attention kernels).
forward(inputs=(query, key, value), attn_mask, key_padding_mask, need_weights): sequence-first (L, B, D) tensors, the
packed in_proj_weight (3D, D) is applied slice by slice (:1010-1020), q scaled by head_dim^-0.5 before q k^T (:1022),
additive attn_mask (:1038-1039), softmax (:1041), @ v, out_proj (:1048-1051), weights averaged over the heads (:1054-1060;
key_padding_mask is accepted and ignored, as in the reference).  The four projections are implicit-GEMM launches with
the bias in the epilogue, the core is tlxmi_mha (any query / key lengths: the decoder's cross attention has 100 queries
over H*W/32^2 memory tokens).  The rest of DETR (backbone with FrozenBatchNorm, transformer stacks, Hungarian matcher,
losses) is out of scope."""
from ... import engine as E
from ...tlx import nn

__all__ = ["MultiHeadAttention"]


class MultiHeadAttention(nn.Module):
    def __init__(self, model_dim, num_heads, dropout=0.0, name="multihead_attn"):
        super().__init__(name=name)
        self.model_dim, self.num_heads = model_dim, num_heads
        assert model_dim % num_heads == 0
        self.head_dim = model_dim // num_heads
        self.dropout = nn.Dropout(dropout)
        in_dim = self.model_dim * 3
        xu = self.str_to_init("xavier_uniform")
        self.in_proj_weight = self._get_weights(var_name="in_proj_weight", shape=(in_dim, self.model_dim), init=xu, trainable=True)
        self.in_proj_bias = self._get_weights(var_name="in_proj_bias", shape=(in_dim,), init=xu, trainable=True)
        self.out_proj_weight = self._get_weights(var_name="out_proj_weight", shape=(self.model_dim, self.model_dim), init=xu, trainable=True)
        self.out_proj_bias = self._get_weights(var_name="out_proj_bias", shape=(self.model_dim,), init=xu, trainable=True)

    def _proj(self, x, part):
        dt = E.precision()
        D = self.model_dim
        if part == "out":
            w, b = self.out_proj_weight, self.out_proj_bias
        else:
            i = "qkv".index(part)
            w, b = self.in_proj_weight[i * D:(i + 1) * D], self.in_proj_bias[i * D:(i + 1) * D]      # detr.py:1010-1020
        pk = self._cached(("pk", part), lambda: E.PackedFilter(w.detach().contiguous(), dt))
        bb = self._cached(("b", part), lambda: E._f32(b.detach().contiguous()))
        return E.linear(x if x.dtype == dt else x.to(dt), pk, bb)

    def forward(self, inputs, attn_mask=None, key_padding_mask=None, need_weights=True):
        query, key, value = inputs
        E.need_gpu(query, "query")
        Q, K, V = self._proj(query, "q"), self._proj(key, "k"), self._proj(value, "v")
        a, w = E.mha(Q, K, V, self.num_heads, float(self.head_dim) ** -0.5, attn_mask, need_weights)
        out = self._proj(a, "out")
        return (out, w) if need_weights else out
