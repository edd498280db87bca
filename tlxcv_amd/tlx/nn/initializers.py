"""`tensorlayerx.nn.initializers` subset used by the reference model files
(vision_transformer.py:17-19, swin_transformer.py:26-28,369).  Initial values never matter for
parity (fixtures assign every weight explicitly); they only have to be finite and reproducible.
"""
import math

import torch

__all__ = ["Constant", "Zeros", "Ones", "TruncatedNormal", "RandomNormal", "RandomUniform", "random_uniform", "xavier_uniform", "XavierUniform",
           "he_normal", "HeNormal", "str_to_init"]

_gen = torch.Generator().manual_seed(0)


class Constant:
    def __init__(self, value=0.0):
        self.value = value

    def __call__(self, shape, dtype=torch.float32):
        return torch.full(tuple(shape), float(self.value), dtype=torch.float32)


class Zeros(Constant):
    def __init__(self):
        super().__init__(0.0)


class Ones(Constant):
    def __init__(self):
        super().__init__(1.0)


class RandomNormal:
    def __init__(self, mean=0.0, stddev=0.05, seed=None):
        self.mean, self.stddev = mean, stddev

    def __call__(self, shape, dtype=torch.float32):
        return torch.randn(tuple(shape), generator=_gen) * self.stddev + self.mean


class RandomUniform:
    """random_uniform(minval, maxval) — alexnet.py:6,32-33,134."""

    def __init__(self, minval=-0.05, maxval=0.05, seed=None):
        self.minval, self.maxval = minval, maxval

    def __call__(self, shape, dtype=torch.float32):
        return torch.rand(tuple(shape), generator=_gen) * (self.maxval - self.minval) + self.minval


random_uniform = RandomUniform


class TruncatedNormal:
    def __init__(self, mean=0.0, stddev=0.05, seed=None):
        self.mean, self.stddev = mean, stddev

    def __call__(self, shape, dtype=torch.float32):
        t = torch.randn(tuple(shape), generator=_gen).clamp_(-2.0, 2.0)
        return t * self.stddev + self.mean


def _fans(shape):
    shape = tuple(shape)
    if len(shape) == 2:
        return shape[0], shape[1]
    rf = 1
    for s in shape[2:]:
        rf *= s
    return shape[1] * rf, shape[0] * rf


class XavierUniform:
    def __init__(self, gain=1.0, seed=None):
        self.gain = gain

    def __call__(self, shape, dtype=torch.float32):
        fi, fo = _fans(shape) if len(tuple(shape)) > 1 else (tuple(shape)[0], tuple(shape)[0])
        lim = self.gain * math.sqrt(6.0 / (fi + fo))
        return (torch.rand(tuple(shape), generator=_gen) * 2 - 1) * lim


class HeNormal:
    def __init__(self, a=0, mode="fan_in", nonlinearity="relu", seed=None):
        pass

    def __call__(self, shape, dtype=torch.float32):
        fi, _ = _fans(shape)
        return torch.randn(tuple(shape), generator=_gen) * math.sqrt(2.0 / max(fi, 1))


xavier_uniform = XavierUniform
he_normal = HeNormal

_BY_NAME = {
    "constant": lambda: Constant(0.0),
    "zeros": Zeros,
    "ones": Ones,
    "truncated_normal": lambda: TruncatedNormal(stddev=0.02),
    "random_normal": RandomNormal,
    "random_uniform": RandomUniform,
    "xavier_uniform": XavierUniform,
    "he_normal": HeNormal,
}


def str_to_init(s):
    if callable(s):
        return s
    if s is None:
        return Constant(0.0)
    try:
        return _BY_NAME[s]()
    except KeyError:
        raise ValueError(f"unknown initializer {s!r}") from None
