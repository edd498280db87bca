"""torch-CPU layers behind the tensorlayerx.nn names (oracle-side stand-in)."""
import torch
import torch.nn.functional as F

from . import initializers  # noqa: F401
from .initializers import Constant, TruncatedNormal, xavier_uniform, str_to_init  # noqa: F401


def Parameter(data=None, name=None):
    return torch.nn.Parameter(torch.as_tensor(data, dtype=torch.float32).detach().clone(), requires_grad=False)


def _falsy_bias(b_init):
    return b_init is None or b_init is False or (isinstance(b_init, (tuple, list)) and len(b_init) == 0)


def str_to_act(act):
    """`act` given by name (resnext.py:46-52 BatchNorm(act='relu'))."""
    if act is None or not isinstance(act, str):
        return act
    return {"relu": F.relu, "relu6": F.relu6, "sigmoid": torch.sigmoid, "gelu": F.gelu, "hardswish": F.hardswish,
            "hard_sigmoid": F.hardsigmoid, "leaky_relu": F.leaky_relu, "tanh": torch.tanh}[act.lower()]


def _tup2(v):
    return (v, v) if isinstance(v, int) else tuple(int(a) for a in v)


class Module(torch.nn.Module):
    def __init__(self, name=None, act=None, *a, **k):
        super().__init__()
        self.name = name
        self.is_train = True

    def _adopt_lists(self):
        seen, stack = set(), [self]
        while stack:
            m = stack.pop()
            if id(m) in seen:
                continue
            seen.add(id(m))
            for k, v in list(vars(m).items()):
                if isinstance(v, (list, tuple)) and v and all(isinstance(e, torch.nn.Module) for e in v):
                    for i, e in enumerate(v):
                        if f"{k}_{i}" not in m._modules and not any(e is r for r in m._modules.values()):
                            m.add_module(f"{k}_{i}", e)
            stack.extend(m._modules.values())

    def state_dict(self, *args, **kwargs):
        self._adopt_lists()
        return super().state_dict(*args, **kwargs)

    def set_eval(self):
        self._adopt_lists()
        for m in self.modules():
            m.training = False
            if isinstance(m, Module):
                m.is_train = False
        return self

    def set_train(self):
        raise NotImplementedError("oracle stand-in is eval-only")

    def register_parameter(self, name=None, param=None):
        return super().register_parameter(name, param)

    def _get_weights(self, var_name, shape, init=None, trainable=True, order=False):
        init = str_to_init(init)
        p = Parameter(data=init(shape=tuple(shape)))
        if trainable:
            self.register_parameter(var_name, p)
            return p
        self.register_buffer(var_name, p.data)
        return getattr(self, var_name)

    def load_dict(self, named, strict=True):
        self._adopt_lists()
        sd = self.state_dict()
        unknown = [k for k in named if k not in sd]
        missing = [k for k in sd if k not in named and sd[k].is_floating_point() and not k.endswith(("attn_mask",))]
        if strict and (unknown or missing):
            raise KeyError(f"load_dict: unknown={unknown[:4]} missing={missing[:4]}")
        with torch.no_grad():
            for k, v in named.items():
                if k in sd:
                    sd[k].copy_(torch.as_tensor(v).to(sd[k].dtype))


class Identity(Module):
    def forward(self, x):
        return x


class Sequential(Module):
    def __init__(self, *layers, name=None):
        super().__init__(name=name)
        if len(layers) == 1 and isinstance(layers[0], (list, tuple)):
            layers = layers[0]
        for i, l in enumerate(layers):
            self.add_module(str(i), l)

    def __iter__(self):
        return iter(self._modules.values())

    def __len__(self):
        return len(self._modules)

    def __getitem__(self, i):
        return list(self._modules.values())[i]

    def forward(self, x):
        for l in self._modules.values():
            x = l(x)
        return x


class ModuleList(torch.nn.ModuleList):
    def __init__(self, modules=None, name=None):
        super().__init__(modules)


def _nchw(x, data_format):
    return x if data_format == "channels_first" else x.permute(0, 3, 1, 2)


def _back(y, data_format):
    return y if data_format == "channels_first" else y.permute(0, 2, 3, 1)


class GroupConv2d(Module):
    def __init__(self, out_channels=32, kernel_size=(1, 1), stride=(1, 1), n_group=1, act=None, padding="SAME",
                 data_format="channels_last", dilation=(1, 1), W_init="truncated_normal", b_init="constant",
                 in_channels=None, name=None):
        super().__init__(name=name)
        self.kernel_size, self.stride, self.dilation = _tup2(kernel_size), _tup2(stride), _tup2(dilation)
        self.n_group, self.data_format, self.act = n_group, data_format, str_to_act(act)
        if isinstance(padding, str):
            self.padding = (0, 0) if padding.upper() == "VALID" else tuple(
                d * (k - 1) // 2 for k, d in zip(self.kernel_size, self.dilation))
        else:
            self.padding = _tup2(padding)
        self.filters = Parameter(str_to_init(W_init)(shape=(out_channels, in_channels // n_group) + self.kernel_size))
        self.biases = None if _falsy_bias(b_init) else Parameter(torch.zeros(out_channels))

    def forward(self, x):
        y = F.conv2d(_nchw(x, self.data_format), self.filters, self.biases, self.stride, self.padding, self.dilation,
                     self.n_group)
        y = _back(y, self.data_format)
        return self.act(y) if self.act is not None else y


class Conv2d(GroupConv2d):
    def __init__(self, out_channels=32, kernel_size=(3, 3), stride=(1, 1), act=None, padding="SAME",
                 data_format="channels_last", dilation=(1, 1), W_init="truncated_normal", b_init="constant",
                 in_channels=None, name=None):
        super().__init__(out_channels, kernel_size, stride, 1, act, padding, data_format, dilation, W_init, b_init,
                         in_channels, name)


class BatchNorm2d(Module):
    def __init__(self, momentum=0.9, epsilon=1e-5, act=None, is_train=True, beta_init="zeros", gamma_init="ones",
                 moving_mean_init="zeros", moving_var_init="ones", num_features=None, data_format="channels_last",
                 name=None):
        super().__init__(name=name)
        self.epsilon, self.data_format, self.act = epsilon, data_format, str_to_act(act)
        n = (num_features,)
        self.gamma = Parameter(str_to_init(gamma_init)(shape=n))
        self.beta = Parameter(str_to_init(beta_init)(shape=n))
        self.register_buffer("moving_mean", str_to_init(moving_mean_init)(shape=n))
        self.register_buffer("moving_var", str_to_init(moving_var_init)(shape=n))

    def forward(self, x):
        assert not self.is_train, "oracle stand-in is eval-only: call set_eval()"
        y = F.batch_norm(_nchw(x, self.data_format), self.moving_mean, self.moving_var, self.gamma, self.beta, False,
                         0.0, self.epsilon)
        y = _back(y, self.data_format)
        return self.act(y) if self.act is not None else y


BatchNorm = BatchNorm2d


class LayerNorm(Module):
    def __init__(self, normalized_shape, epsilon=1e-5, gamma_init="ones", beta_init="zeros", act=None, name=None):
        super().__init__(name=name)
        if isinstance(normalized_shape, int):
            normalized_shape = (normalized_shape,)
        self.normalized_shape, self.epsilon = tuple(normalized_shape), epsilon
        self.gamma = Parameter(torch.ones(self.normalized_shape))
        self.beta = Parameter(torch.zeros(self.normalized_shape))

    def forward(self, x):
        return F.layer_norm(x, self.normalized_shape, self.gamma, self.beta, self.epsilon)


class Linear(Module):
    def __init__(self, out_features=None, act=None, W_init="truncated_normal", b_init="constant", in_features=None,
                 name=None):
        super().__init__(name=name)
        self.act = act
        self.weights = Parameter(str_to_init(W_init)(shape=(in_features, out_features)))
        self.biases = None if _falsy_bias(b_init) else Parameter(torch.zeros(out_features))

    def forward(self, x):
        y = torch.matmul(x, self.weights)
        if self.biases is not None:
            y = y + self.biases
        return self.act(y) if self.act is not None else y


class MaxPool2d(Module):
    def __init__(self, kernel_size, stride=None, padding="SAME", return_mask=False, data_format="channels_last",
                 name=None):
        super().__init__(name=name)
        self.kernel_size = _tup2(kernel_size)
        self.stride = _tup2(stride if stride is not None else kernel_size)
        self.padding = _tup2(padding) if not isinstance(padding, str) else (
            (0, 0) if padding.upper() == "VALID" else tuple((k - 1) // 2 for k in self.kernel_size))
        self.data_format = data_format

    def forward(self, x):
        return _back(F.max_pool2d(_nchw(x, self.data_format), self.kernel_size, self.stride, self.padding),
                     self.data_format)


class AdaptiveAvgPool2d(Module):
    def __init__(self, output_size, data_format="channels_last", name=None):
        super().__init__(name=name)
        self.output_size, self.data_format = output_size, data_format

    def forward(self, x):
        return _back(F.adaptive_avg_pool2d(_nchw(x, self.data_format), self.output_size), self.data_format)


class AdaptiveAvgPool1d(Module):
    def __init__(self, output_size, data_format="channels_first", name=None):
        super().__init__(name=name)
        self.output_size = output_size

    def forward(self, x):
        return F.adaptive_avg_pool1d(x, self.output_size)


class Dropout(Module):
    def __init__(self, p=0.5, seed=0, name=None):
        super().__init__(name=name)

    def forward(self, x):
        return x


class _A(Module):
    fn = staticmethod(lambda x: x)

    def forward(self, x):
        return self.fn(x)


class ReLU(_A):
    fn = staticmethod(F.relu)


class ReLU6(_A):
    fn = staticmethod(F.relu6)


class Hardswish(_A):
    fn = staticmethod(F.hardswish)


class HardSigmoid(_A):
    fn = staticmethod(F.hardsigmoid)


class Sigmoid(_A):
    fn = staticmethod(torch.sigmoid)


class GELU(_A):
    fn = staticmethod(lambda x: F.gelu(x, approximate="none"))


class LeakyReLU(Module):
    def __init__(self, negative_slope=0.01, name=None):
        super().__init__(name=name)
        self.negative_slope = negative_slope

    def forward(self, x):
        return F.leaky_relu(x, self.negative_slope)


class Softmax(Module):
    def __init__(self, axis=-1, name=None):
        super().__init__(name=name)
        self.axis = axis

    def forward(self, x):
        return torch.softmax(x, dim=self.axis)
