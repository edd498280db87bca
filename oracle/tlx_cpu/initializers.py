"""Initializers for the oracle stand-in (values never matter: fixtures overwrite every weight)."""
import torch

_g = torch.Generator().manual_seed(0)


class Constant:
    def __init__(self, value=0.0):
        self.value = value

    def __call__(self, shape, dtype=None):
        return torch.full(tuple(shape), float(self.value))


class TruncatedNormal:
    def __init__(self, mean=0.0, stddev=0.05, seed=None):
        self.mean, self.stddev = mean, stddev

    def __call__(self, shape, dtype=None):
        return torch.randn(tuple(shape), generator=_g).clamp_(-2, 2) * self.stddev + self.mean


class xavier_uniform:
    def __init__(self, gain=1.0, seed=None):
        pass

    def __call__(self, shape, dtype=None):
        return (torch.rand(tuple(shape), generator=_g) - 0.5) * 0.1


class random_uniform:
    def __init__(self, minval=-0.05, maxval=0.05, seed=None):
        self.minval, self.maxval = minval, maxval

    def __call__(self, shape, dtype=None):
        return torch.rand(tuple(shape), generator=_g) * (self.maxval - self.minval) + self.minval


class he_normal:
    def __init__(self, a=0, mode="fan_in", nonlinearity="leaky_relu", seed=None):
        pass

    def __call__(self, shape, dtype=None):
        return torch.randn(tuple(shape), generator=_g) * 0.05


def str_to_init(s):
    if callable(s):
        return s
    return {"zeros": Constant(0.0), "ones": Constant(1.0), "constant": Constant(0.0), None: Constant(0.0),
            "truncated_normal": TruncatedNormal(stddev=0.02), "xavier_uniform": xavier_uniform(), "he_normal": he_normal()}[s]
