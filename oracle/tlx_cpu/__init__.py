"""Minimal torch-CPU stand-in for `tensorlayerx` (oracle-side only; see oracle/__init__.py).

Layer semantics [TLX-recalled, unverifiable here]: b_init falsy => no bias; int padding = symmetric
zero padding; MaxPool2d pads -inf; BatchNorm eps 1e-5 (eval: moving stats); LayerNorm eps 1e-5,
biased variance; GELU exact erf; Linear weights stored (in, out); conv filters OIHW.
Parameter attribute names match tlxcv_amd.tlx.nn so one seeded dictionary fits both.
"""
import numpy as np
import torch

from . import nn, ops  # noqa: F401
from .nn import initializers  # noqa: F401
from .ops import GeLU, softmax, sigmoid, relu, arange, stack  # noqa: F401

BACKEND = "torch"
float32 = torch.float32
int64 = torch.int64


def convert_to_tensor(value, dtype=None, device=None):
    t = value if isinstance(value, torch.Tensor) else torch.as_tensor(np.asarray(value))
    return t.to(dtype) if dtype is not None else t


def convert_to_numpy(value):
    return value.detach().cpu().numpy()


def get_tensor_shape(x):
    return list(x.shape)


def transpose(a, perm=None, conjugate=False):
    return a.permute(*perm) if perm is not None else a.permute(*reversed(range(a.dim())))


def reshape(tensor, shape):
    return tensor.reshape(tuple(shape))


def flatten(x, start_axis=0, stop_axis=-1):
    return torch.flatten(x, start_axis, stop_axis)


def concat(values, axis=0):
    return torch.cat(list(values), dim=axis)


def split(value, num_or_size_splits, axis=0):
    if isinstance(num_or_size_splits, int):
        return torch.chunk(value, num_or_size_splits, dim=axis)
    return torch.split(value, list(num_or_size_splits), dim=axis)


def expand_dims(input, axis):
    return input.unsqueeze(axis)


def roll(input, shifts, dims=None):
    return torch.roll(input, shifts, dims)


def index_select(x, index, axis=0):
    return torch.index_select(x, axis, index)


def matmul(a, b, transpose_a=False, transpose_b=False):
    if transpose_a:
        a = a.transpose(-1, -2)
    if transpose_b:
        b = b.transpose(-1, -2)
    return torch.matmul(a, b)


def add_n(inputs):
    out = inputs[0]
    for t in inputs[1:]:
        out = out + t
    return out


def multiply(x, y):
    return x * y


def add(value, bias):
    return value + bias


def cast(x, dtype):
    return x.to(dtype)


def floor(x):
    return torch.floor(x)


def zeros(shape, dtype=torch.float32, device=None):
    return torch.zeros(tuple(shape), dtype=dtype)


def ones(shape, dtype=torch.float32, device=None):
    return torch.ones(tuple(shape), dtype=dtype)


def ones_like(x):
    return torch.ones_like(x)


def meshgrid(*args, indexing="ij"):
    if len(args) == 1 and isinstance(args[0], (list, tuple)):
        args = tuple(args[0])
    return torch.meshgrid(*args, indexing=indexing)


def reduce_mean(input_tensor, axis=None, keepdims=False):
    return torch.mean(input_tensor) if axis is None else torch.mean(input_tensor, dim=axis, keepdim=keepdims)


def reduce_max(input_tensor, axis=None, keepdims=False):
    return torch.max(input_tensor) if axis is None else torch.max(input_tensor, dim=axis, keepdim=keepdims).values


def squeeze(input, axis=None):
    return input.squeeze() if axis is None else input.squeeze(axis)


def argsort(values, axis=-1, descending=False):
    return torch.argsort(values, dim=axis, descending=descending, stable=True)


def argmax(x, axis=None, dtype="int64"):
    return torch.argmax(x, dim=axis)


class FlattenReshape(nn.Module):
    def forward(self, x):
        return x.reshape(x.shape[0], -1)
