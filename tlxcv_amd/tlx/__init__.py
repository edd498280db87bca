"""`tensorlayerx`-compatible top level (`import tlxcv_amd.tlx as tlx`), torch backend, MI355X only.

Only what the hot-path model files and the inference demos touch: tensor
functions are thin views/reshapes on torch tensors (no arithmetic), `argmax` and the activations
run HIP kernels.  `tlxcv_amd.install()` makes `import tensorlayerx` resolve to this package.
"""
import numpy as np
import torch

from .. import engine as _E
from . import nn, ops, vision  # noqa: F401
from .nn import initializers  # noqa: F401
from .ops import (GeLU, softmax, sigmoid, relu, arange, stack)  # noqa: F401

BACKEND = "torch"
float32 = torch.float32
float16 = torch.float16
int64 = torch.int64
int32 = torch.int32


def set_device(device="GPU", id=0):
    if str(device).upper() not in ("GPU", "CUDA"):
        raise RuntimeError("tlxcv_amd runs on MI355X only; set_device('GPU', id)")
    torch.cuda.set_device(id)


def convert_to_tensor(value, dtype=None, device=None):
    t = value if isinstance(value, torch.Tensor) else torch.as_tensor(np.asarray(value))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))


def convert_to_numpy(value):
    return value.detach().float().cpu().numpy() if value.is_floating_point() else value.detach().cpu().numpy()


def get_tensor_shape(x):
    return list(x.shape)


def transpose(a, perm=None, conjugate=False):
    if perm is None:
        perm = tuple(reversed(range(a.dim())))
    return a.permute(*perm)


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


def squeeze(input, axis=None):
    return input.squeeze() if axis is None else input.squeeze(axis)


def roll(input, shifts, dims=None):
    return torch.roll(input, shifts, dims)


def index_select(x, index, axis=0):
    return torch.index_select(x, axis, index)


def matmul(a, b, transpose_a=False, transpose_b=False):
    """tlx.matmul (detr.py:1013): on the device every product runs on libtlxmi (engine.matmul), never a BLAS library."""
    return _E.matmul(a, b, transpose_a, transpose_b)


def add(value, bias):
    return value + bias


def cast(x, dtype):
    return x.to(dtype)


def floor(x):
    return torch.floor(x)


def zeros(shape, dtype=torch.float32, device=None):
    return torch.zeros(tuple(shape), dtype=dtype, device=device)


def ones(shape, dtype=torch.float32, device=None):
    return torch.ones(tuple(shape), dtype=dtype, device=device)


def ones_like(x):
    return torch.ones_like(x)


def meshgrid(*args, indexing="ij"):
    if len(args) == 1 and isinstance(args[0], (list, tuple)):
        args = tuple(args[0])
    return torch.meshgrid(*args, indexing=indexing)


def argmax(x, axis=None, dtype="int64"):
    """tlx.argmax(outputs, axis=-1) — tasks/image_classification.py:23; runs tlxmi_argmax_lastdim."""
    _E.need_gpu(x, "logits")
    if axis is None:
        x, axis = x.reshape(-1), -1
    if axis not in (-1, x.dim() - 1):
        x = x.transpose(axis, -1)
    if x.dtype not in (torch.float16, torch.float32):
        x = x.float()
    return _E.argmax_lastdim(x)


class FlattenReshape(nn.Flatten):
    """tlx.FlattenReshape() — resnet.py:232."""
