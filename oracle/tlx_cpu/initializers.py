"""Initializers for the oracle stand-in (values never matter: fixtures overwrite every weight).
Called with a shape they return a tensor; called with a tensor (the Paddle spelling, swin_transformer.py:166
`trunc_normal_(table)`) they fill it in place."""
import functools

import torch

_g = torch.Generator().manual_seed(0)


def _inplace_ok(call):
    @functools.wraps(call)
    def run(self, shape=None, dtype=None):
        if isinstance(shape, torch.Tensor):
            with torch.no_grad():
                shape.copy_(call(self, tuple(shape.shape)))
            return shape
        return call(self, shape)
    return run


class Constant:
    def __init__(self, value=0.0):
        self.value = value

    @_inplace_ok
    def __call__(self, shape, dtype=None):
        return torch.full(tuple(shape), float(self.value))


class TruncatedNormal:
    def __init__(self, mean=0.0, stddev=0.05, seed=None):
        self.mean, self.stddev = mean, stddev

    @_inplace_ok
    def __call__(self, shape, dtype=None):
        return torch.randn(tuple(shape), generator=_g).clamp_(-2, 2) * self.stddev + self.mean


class xavier_uniform:
    def __init__(self, gain=1.0, seed=None):
        pass

    @_inplace_ok
    def __call__(self, shape, dtype=None):
        return (torch.rand(tuple(shape), generator=_g) - 0.5) * 0.1


class random_uniform:
    def __init__(self, minval=-0.05, maxval=0.05, seed=None):
        self.minval, self.maxval = minval, maxval

    @_inplace_ok
    def __call__(self, shape, dtype=None):
        return torch.rand(tuple(shape), generator=_g) * (self.maxval - self.minval) + self.minval


class he_normal:
    def __init__(self, a=0, mode="fan_in", nonlinearity="leaky_relu", seed=None):
        pass

    @_inplace_ok
    def __call__(self, shape, dtype=None):
        return torch.randn(tuple(shape), generator=_g) * 0.05


def str_to_init(s):
    if callable(s):
        return s
    return {"zeros": Constant(0.0), "ones": Constant(1.0), "constant": Constant(0.0), None: Constant(0.0),
            "truncated_normal": TruncatedNormal(stddev=0.02), "xavier_uniform": xavier_uniform(), "he_normal": he_normal()}[s]
