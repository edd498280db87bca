import torch

from .nn import GELU as GeLU  # noqa: F401


def softmax(logits, axis=-1):
    return torch.softmax(logits, dim=axis)


def sigmoid(x):
    return torch.sigmoid(x)


def relu(x):
    return torch.relu(x)


def arange(start, limit=None, delta=1, dtype=None):
    return torch.arange(start, limit, delta, dtype=dtype) if limit is not None else torch.arange(start, dtype=dtype)


def stack(values, axis=0):
    return torch.stack(list(values), dim=axis)


def random_uniform(shape, minval=0, maxval=1, dtype=torch.float32, seed=None):
    return torch.rand(tuple(shape)) * (maxval - minval) + minval
