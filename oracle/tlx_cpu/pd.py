"""Paddle tensor-method semantics for the reference's Paddle-converted model files (development container only).

swin_transformer.py calls Paddle methods on its tensors — `x.transpose([0, 1, 3, 2, 4, 5])` (:97), `x.unsqueeze(axis=2)`
(:150), `(mask != 0).astype('float32')` (:306) — which torch tensors do not take.  PdTensor is a torch.Tensor subclass
that adds exactly those spellings; torch propagates the subclass through every op, so a reference file loaded unmodified
runs on torch-CPU arithmetic.  `pd_module` wraps a module's functions so tensors they create come back as PdTensor."""
import types

import torch

_DTYPES = {"float32": torch.float32, "float64": torch.float64, "int64": torch.int64, "int32": torch.int32,
           "bool": torch.bool, "float16": torch.float16}


class PdTensor(torch.Tensor):
    def transpose(self, *perm, **kw):
        if "perm" in kw:
            perm = (kw["perm"],)
        if len(perm) == 1 and isinstance(perm[0], (list, tuple)):
            return self.permute(*perm[0])
        return super().transpose(*perm)

    def unsqueeze(self, axis=None, dim=None):
        return super().unsqueeze(axis if axis is not None else dim)

    def squeeze(self, axis=None, dim=None):
        a = axis if axis is not None else dim
        return super().squeeze() if a is None else super().squeeze(a)

    def astype(self, dtype):
        return self.to(_DTYPES[dtype] if isinstance(dtype, str) else dtype)

    def numpy(self):
        return self.as_subclass(torch.Tensor).detach().numpy()


def wrap(v):
    if isinstance(v, torch.nn.Parameter):
        return v
    if isinstance(v, torch.Tensor):
        return v.as_subclass(PdTensor)
    if isinstance(v, (list, tuple)):
        return type(v)(wrap(e) for e in v)
    return v


def unwrap(v):
    return v.as_subclass(torch.Tensor) if isinstance(v, PdTensor) else v


def _wrapping(fn):
    def call(*a, **k):
        return wrap(fn(*a, **k))
    call.__name__ = getattr(fn, "__name__", "fn")
    call.__doc__ = fn.__doc__
    return call


def pd_module(mod, name=None, **replace):
    """A copy of module `mod` whose plain functions return PdTensor; classes and sub-modules are shared."""
    out = types.ModuleType(name or mod.__name__)
    for k, v in vars(mod).items():
        if k.startswith("__") and k not in ("__doc__",):
            continue
        out.__dict__[k] = _wrapping(v) if isinstance(v, types.FunctionType) else v
    out.__dict__.update(replace)
    return out
