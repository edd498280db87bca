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

    def str_to_init(self, s):
        return str_to_init(s)

    def _get_weights(self, var_name, shape, init=None, trainable=True, order=False):
        init = str_to_init(init)
        p = Parameter(data=init(shape=tuple(shape)))
        if trainable:
            self.register_parameter(var_name, p)
            return p
        self.register_buffer(var_name, p.data)
        return getattr(self, var_name)

    # Paddle-backend Layer methods the converted files call (swin_transformer.py:141-145,163-165)
    def create_parameter(self, shape, attr=None, dtype=None, is_bias=False, default_initializer=None):
        init = default_initializer if default_initializer is not None else Constant(0.0)
        return Parameter(data=init(shape=tuple(shape)))

    def add_parameter(self, name, parameter):
        if name in self.__dict__:
            del self.__dict__[name]
        self.register_parameter(name, parameter)
        return parameter

    # --- checkpoint interchange [TLX-recalled; the real package cannot be consulted here]: `all_weights` lists the
    # variables layer by layer in construction order, each layer's in the order its build() creates them (conv / linear:
    # weight, bias; BatchNorm: beta, gamma, moving_mean, moving_var; LayerNorm: gamma, beta); save_weights('x.npz') writes
    # that list, without names, as ONE object array under the key `params` (tlx.files.save_npz) and load_weights assigns
    # it back by position (demo/image_classification/train.py:55, predict.py:19).
    _BUILD_ORDER = {"BatchNorm2d": ("beta", "gamma", "moving_mean", "moving_var")}

    @property
    def all_weights(self):
        self._adopt_lists()
        out = []
        for _, m in self.named_modules():
            own = dict(m._parameters)
            own.update(m._buffers)
            for k in Module._BUILD_ORDER.get(type(m).__name__, list(own)):
                if own.get(k) is not None and k not in ("attn_mask", "relative_position_bias", "relative_position_index"):
                    out.append(own[k])
        return out

    def save_weights(self, file_path, format=None):
        import numpy as np
        assert (format or "npz") == "npz", "stand-in writes the positional npz only"
        ws = self.all_weights
        params = np.empty(len(ws), dtype=object)
        for i, w in enumerate(ws):
            params[i] = w.detach().numpy().astype(np.float32)
        np.savez(file_path, params=params)

    def load_weights(self, file_path, format=None, in_order=True, skip=False):
        import numpy as np
        params = list(np.load(file_path, allow_pickle=True)["params"])
        ws = self.all_weights
        assert len(params) == len(ws)
        with torch.no_grad():
            for w, a in zip(ws, params):
                w.copy_(torch.as_tensor(a))

    def load_dict(self, named, strict=True):
        self._adopt_lists()
        sd = self.state_dict()
        unknown = [k for k in named if k not in sd]
        missing = [k for k in sd if k not in named and sd[k].is_floating_point()
                   and not k.endswith(("attn_mask", "relative_position_bias"))]
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
        if len(layers) == 1 and isinstance(layers[0], dict):     # OrderedDict of named layers (resnest.py:479-513)
            for k, l in layers[0].items():
                self.add_module(k, l)
            return
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
    """padding='SAME' follows TensorLayerX's torch backend [TLX-recalled]: TensorFlow's rule — total padding
    max(0, (ceil(in/stride) - 1) * stride + dilation * (k - 1) + 1 - in), the odd unit on the bottom / right — which for
    odd kernels at stride 1 is the symmetric (k - 1) // 2.  in_channels=None defers the filter to the first call
    (efficientnet.py:92-125 builds every layer that way and runs one forward at construction, :433-441)."""

    def __init__(self, out_channels=32, kernel_size=(1, 1), stride=(1, 1), n_group=1, act=None, padding="SAME",
                 data_format="channels_last", dilation=(1, 1), W_init="truncated_normal", b_init="constant",
                 in_channels=None, name=None):
        super().__init__(name=name)
        self.kernel_size, self.stride, self.dilation = _tup2(kernel_size), _tup2(stride), _tup2(dilation)
        self.n_group, self.data_format, self.act = n_group, data_format, str_to_act(act)
        self.out_channels = out_channels
        self.same = False
        if isinstance(padding, str):
            self.same = padding.upper() == "SAME"
            self.padding = (0, 0)
        else:
            self.padding = _tup2(padding)
        self._w_init, self._has_bias = W_init, not _falsy_bias(b_init)
        self.filters = self.biases = None
        if in_channels is not None:
            self._build(in_channels)

    def _build(self, in_channels):
        self.filters = Parameter(str_to_init(self._w_init)(shape=(self.out_channels, in_channels // self.n_group) + self.kernel_size))
        self.biases = Parameter(torch.zeros(self.out_channels)) if self._has_bias else None

    def forward(self, x):
        x = _nchw(x, self.data_format)
        if self.filters is None:
            self._build(x.shape[1])
        pad = self.padding
        if self.same:
            lo = []
            hi = []
            for i, k, s_, d in zip(x.shape[2:], self.kernel_size, self.stride, self.dilation):
                total = max(0, (-(-i // s_) - 1) * s_ + d * (k - 1) + 1 - i)
                lo.append(total // 2)
                hi.append(total - total // 2)
            x = F.pad(x, (lo[1], hi[1], lo[0], hi[0]))
            pad = (0, 0)
        y = F.conv2d(x, self.filters, self.biases, self.stride, pad, self.dilation, self.n_group)
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
        self._inits = (gamma_init, beta_init, moving_mean_init, moving_var_init)
        self.gamma = None
        if num_features is not None:
            self._build(num_features)

    def _build(self, num_features):
        n = (num_features,)
        gi, bi, mi, vi = self._inits
        self.gamma = Parameter(str_to_init(gi)(shape=n))
        self.beta = Parameter(str_to_init(bi)(shape=n))
        self.register_buffer("moving_mean", str_to_init(mi)(shape=n))
        self.register_buffer("moving_var", str_to_init(vi)(shape=n))

    def forward(self, x):
        xn = _nchw(x, self.data_format)
        if self.gamma is None:          # deferred build (efficientnet.py:433-441): shapes only, the values are never used
            self._build(xn.shape[1])
            return x
        assert not self.is_train, "oracle stand-in is eval-only: call set_eval()"
        y = F.batch_norm(xn, self.moving_mean, self.moving_var, self.gamma, self.beta, False, 0.0, self.epsilon)
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
        self.act = str_to_act(act)
        self.out_features, self._w_init, self._has_bias = out_features, W_init, not _falsy_bias(b_init)
        self.weights = self.biases = None
        if in_features is not None:
            self._build(in_features)

    def _build(self, in_features):
        self.weights = Parameter(str_to_init(self._w_init)(shape=(in_features, self.out_features)))
        self.biases = Parameter(torch.zeros(self.out_features)) if self._has_bias else None

    def forward(self, x):
        if self.weights is None:
            self._build(x.shape[-1])
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


class AvgPool2d(Module):
    """nn.AvgPool2d(kernel_size, stride, padding, data_format) — resnest.py:212-218, 250-256, 271-286.  Zero padding
    counts in the divisor (torch's default, which TensorLayerX's torch backend forwards to) [TLX-recalled]."""

    def __init__(self, kernel_size, stride=None, padding="SAME", ceil_mode=False, data_format="channels_last", name=None):
        super().__init__(name=name)
        self.kernel_size = _tup2(kernel_size)
        self.stride = _tup2(stride if stride is not None else kernel_size)
        self.padding = _tup2(padding) if not isinstance(padding, str) else (
            (0, 0) if padding.upper() == "VALID" else tuple((k - 1) // 2 for k in self.kernel_size))
        self.data_format = data_format

    def forward(self, x):
        return _back(F.avg_pool2d(_nchw(x, self.data_format), self.kernel_size, self.stride, self.padding),
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


class Flatten(Module):
    def forward(self, x):
        return x.reshape(x.shape[0], -1)


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
