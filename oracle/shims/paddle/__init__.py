"""`paddle` names the reference's Paddle-converted model files touch on their eval-mode forward path
(swin_transformer.py:43-45,305; see ../README.md).  Development container only."""
import torch

from oracle.tlx_cpu.pd import PdTensor, wrap

Tensor = PdTensor


def ones_like(x, dtype=None):
    return wrap(torch.ones_like(x))


def zeros_like(x, dtype=None):
    return wrap(torch.zeros_like(x))


def shape(x):
    return list(x.shape)


def rand(shape, dtype=None):
    return wrap(torch.rand(tuple(int(s) for s in shape)))


def to_tensor(data, dtype=None):
    return wrap(torch.as_tensor(data))
