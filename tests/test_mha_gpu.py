"""Row g1 / f4: `tlx.nn.MultiheadAttention` (named by north_star) and DETR's `MultiHeadAttention`
(tlxcv/models/detection/detr.py:965-1062) on the engine, against the fixture written by the reference's own class
(tests/golden/detr_mha.npz: a masked cross-attention case with head-averaged weights and a 197-token self-attention
case) and against the oracle restatement."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import functional as OF
from tlxcv_amd import seeded

pytestmark = pytest.mark.gpu


def _case(g, tag, dev):
    D, H, T, S, B = (int(v) for v in g[f"{tag}_dims"])
    q = torch.from_numpy(g[f"{tag}_q"]).to(dev)
    kv = q if tag == "self" else torch.from_numpy(g[f"{tag}_kv"]).to(dev)
    mask = torch.from_numpy(g[f"{tag}_mask"]).to(dev) if f"{tag}_mask" in g.files else None
    return D, H, q, kv, mask


@pytest.mark.parametrize("tag", ["cross", "self"])
def test_detr_multiheadattention_fp32_matches_the_reference_class(dev, fp32_mode, tag):
    from tlxcv_amd.models import MultiHeadAttention
    g = np.load(os.path.join(GOLDEN, "detr_mha.npz"))
    D, H, q, kv, mask = _case(g, tag, dev)
    m = MultiHeadAttention(D, H)
    assert list(seeded.shapes_of(m)) == [str(k) for k in g["param_names"]]
    m.load_dict(seeded.fill(seeded.shapes_of(m), int(g["weight_seed"])))
    m = m.to(dev).set_eval()
    out, w = m((q, kv, kv), attn_mask=mask)
    assert np.abs(out.cpu().numpy() - g[f"{tag}_out"]).max() <= 1e-4
    assert np.abs(w.cpu().numpy() - g[f"{tag}_weights"]).max() <= 1e-5
    only = m((q, kv, kv), attn_mask=mask, need_weights=False)          # detr.py:1062: the bare tensor
    assert isinstance(only, torch.Tensor) and np.abs(only.cpu().numpy() - g[f"{tag}_out"]).max() <= 1e-4
    # a per-(batch, head) mask of the same values gives the same result (detr.py:1038: broadcast add)
    if mask is not None:
        B = q.shape[1]
        out2, _ = m((q, kv, kv), attn_mask=mask[None].expand(B * H, -1, -1).contiguous())
        assert torch.equal(out2, out)


@pytest.mark.parametrize("tag", ["cross", "self"])
def test_detr_multiheadattention_fp16_tracks_the_reference_class(dev, fp16_mode, tag):
    from tlxcv_amd.models import MultiHeadAttention
    g = np.load(os.path.join(GOLDEN, "detr_mha.npz"))
    D, H, q, kv, mask = _case(g, tag, dev)
    m = MultiHeadAttention(D, H)
    m.load_dict(seeded.fill(seeded.shapes_of(m), int(g["weight_seed"])))
    m = m.to(dev).set_eval()
    out, w = m((q, kv, kv), attn_mask=mask)
    ref = g[f"{tag}_out"]
    assert np.abs(out.float().cpu().numpy() - ref).max() <= 0.003 * (ref.max() - ref.min())
    assert np.abs(w.cpu().numpy() - g[f"{tag}_weights"]).max() <= 2e-3


def _tlx_mha(D, H, dev, seed=5, **kw):
    from tlxcv_amd.tlx import nn
    m = nn.MultiheadAttention(D, H, **kw)
    params = seeded.fill(seeded.shapes_of(m), seed)
    m.load_dict(params)
    # the same weights as DETR's packed in_proj / out_proj (detr.py:1010-1020) for the oracle
    p = {"in_proj_weight": np.concatenate([params["q_weight"], params["k_weight"], params["v_weight"]]),
         "in_proj_bias": np.concatenate([params["q_bias"], params["k_bias"], params["v_bias"]]),
         "out_proj_weight": params["out_weight"], "out_proj_bias": params["out_bias"]}
    return m.to(dev).set_eval(), {k: torch.from_numpy(v) for k, v in p.items()}


def test_tlx_multiheadattention_general_path(dev, fp32_mode):
    rng = np.random.default_rng(3)
    m, p = _tlx_mha(96, 3, dev)
    q = torch.from_numpy(rng.standard_normal((9, 2, 96)).astype(np.float32))
    kv = torch.from_numpy(rng.standard_normal((300, 2, 96)).astype(np.float32))         # > 256 keys: several key tiles
    mask = torch.from_numpy(np.where(rng.random((9, 300)) < 0.3, -np.inf, 0.0).astype(np.float32))
    mask[:, 5] = 0.0
    with torch.no_grad():
        ro, rw = OF.detr_mha(p, "", q, kv, kv, 3, mask)
    o, w = m(q.to(dev), kv.to(dev), kv.to(dev), attn_mask=mask.to(dev))
    assert np.abs(o.cpu().numpy() - ro.numpy()).max() <= 1e-4 and np.abs(w.cpu().numpy() - rw.numpy()).max() <= 1e-5
    # batch_first: same numbers on transposed tensors
    mb, _ = _tlx_mha(96, 3, dev, batch_first=True)
    ob, wb = mb(q.transpose(0, 1).contiguous().to(dev), kv.transpose(0, 1).contiguous().to(dev), attn_mask=mask.to(dev))
    assert np.abs(ob.transpose(0, 1).cpu().numpy() - ro.numpy()).max() <= 1e-4 and torch.equal(wb, w)
    with pytest.raises(NotImplementedError):
        m(q.to(dev), key_padding_mask=torch.zeros(2, 9, dtype=torch.bool, device=dev))


def test_tlx_multiheadattention_self_attention_takes_the_fused_kernel(dev, fp16_mode):
    """fp16 self attention of <= 256 tokens without mask / weights = packed qkv GEMM + the MFMA attention kernel
    (tlxmi_attention, the one ViT uses); must agree with the general kernel and with the oracle."""
    rng = np.random.default_rng(4)
    m, p = _tlx_mha(128, 2, dev, need_weights=False)
    x = torch.from_numpy(rng.standard_normal((197, 3, 128)).astype(np.float32))
    with torch.no_grad():
        ro = OF.detr_mha(p, "", x, x, x, 2, None, need_weights=False)
    o, w = m(x.to(dev))
    assert w is None and tuple(o.shape) == (197, 3, 128)
    rngo = float(ro.max() - ro.min())
    assert np.abs(o.float().cpu().numpy() - ro.numpy()).max() <= 0.003 * rngo
    xk = x.clone().to(dev)                                   # a distinct key tensor: the general kernel
    o2, _ = m(x.to(dev), xk, xk)
    assert (o2.float() - o.float()).abs().max().item() <= 0.003 * rngo
