"""`tensorlayerx.ops` subset (vision_transformer.py:70,118; swin_transformer.py:143-146)."""
import torch

from .. import engine as _E
from .nn import GELU as GeLU  # noqa: F401  tlx.ops.GeLU is used as a layer class (vision_transformer.py:70)


def softmax(logits, axis=-1):
    return _E.softmax(logits, axis)


def sigmoid(x):
    _E.need_gpu(x)
    return _E.act_flat(x, _E.ACT_SIGMOID)


def relu(x):
    _E.need_gpu(x)
    return _E.act_flat(x, _E.ACT_RELU)


def arange(start, limit=None, delta=1, dtype=None):
    return torch.arange(start, limit, delta, dtype=dtype) if limit is not None else torch.arange(start, dtype=dtype)


def stack(values, axis=0):
    return torch.stack(list(values), dim=axis)


def convert_to_tensor(value, dtype=None):
    from . import convert_to_tensor as c
    return c(value, dtype)
