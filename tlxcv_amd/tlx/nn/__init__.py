"""`tensorlayerx.nn`-compatible layer classes whose forward runs on libtlxmi.so (MI355X only).

Constructor signatures, attribute names and error behaviour follow the way the reference model
files use TensorLayerX (call sites are cited per class).  TensorLayerX itself is
not vendored by the reference, so its *internal* semantics are restated from its public
documentation and marked [TLX-recalled] where they cannot be checked here:
  - `b_init` falsy ((), False, None) => no bias; anything else => bias, zero-initialised;
  - integer `padding` = symmetric explicit zero padding; MaxPool2d pads with -inf;
  - BatchNorm2d epsilon 1e-5, LayerNorm epsilon 1e-5 unless given; GELU is the exact erf form;
  - conv filters are kept OIHW, Linear weights (in_features, out_features).
Tensors are real `torch.Tensor`s (reference model code calls torch methods on them directly,
e.g. vision_transformer.py:113-120).  With data_format='channels_first' a layer accepts and
returns logical NCHW tensors; physically they are NHWC (torch channels_last strides), which is
what the HIP kernels read, so chained layers never transpose.
Eval-mode forward only: training (batch statistics, dropout masks, autograd) is out of scope.
"""
import math

import numpy as np
import torch

from ... import engine as E
from . import initializers  # noqa: F401  (re-exported as nn.initializers)
from .initializers import Constant, TruncatedNormal, xavier_uniform, he_normal, str_to_init  # noqa: F401

__all__ = [
    "Module", "Sequential", "ModuleList", "Parameter", "GroupConv2d", "Conv2d", "BatchNorm2d", "BatchNorm",
    "LayerNorm", "Linear", "MaxPool2d", "AdaptiveAvgPool2d", "AdaptiveAvgPool1d", "Dropout", "ReLU", "ReLU6",
    "LeakyReLU", "Hardswish", "HardSigmoid", "Sigmoid", "Softmax", "GELU", "Flatten", "UpSampling2d", "Identity",
    "MultiheadAttention",
]


def Parameter(data=None, name=None, requires_grad=False):
    """nn.Parameter(data=...) as used at vision_transformer.py:295."""
    if not isinstance(data, torch.Tensor):
        data = torch.as_tensor(np.asarray(data), dtype=torch.float32)
    return torch.nn.Parameter(data.detach().to(torch.float32), requires_grad=False)


class Module(torch.nn.Module):
    """TensorLayerX-style module: `name`, `is_train`, set_eval()/set_train(), all_weights,
    load_weights/save_weights, sub-modules kept in plain Python lists are adopted
    (darknet.py:270-297, yolov3.py:210,306)."""

    def __init__(self, name=None, act=None, *args, **kwargs):
        super().__init__()
        self.name = name
        self.is_train = True
        self._engine_cache = {}
        self._weights_epoch = 0

    # -- TLX surface -------------------------------------------------------------------------
    def _adopt_lists(self):
        for m in list(self.modules_shallow()):
            for k, v in list(vars(m).items()):
                if isinstance(v, (list, tuple)) and v and all(isinstance(e, torch.nn.Module) for e in v):
                    for i, e in enumerate(v):
                        key = f"{k}_{i}"
                        # (a list whose members are also attributes — resnext.py:176-189 `bb_i_j` + `block_list` —
                        # keeps the attribute names only)
                        if key not in m._modules and not any(e is r for r in m._modules.values()):
                            m.add_module(key, e)

    def modules_shallow(self):
        seen, stack = set(), [self]
        while stack:
            m = stack.pop()
            if id(m) in seen:
                continue
            seen.add(id(m))
            yield m
            stack.extend(m._modules.values())
            for v in vars(m).values():
                if isinstance(v, (list, tuple)):
                    stack.extend(e for e in v if isinstance(e, torch.nn.Module))

    def state_dict(self, *args, **kwargs):
        self._adopt_lists()
        return super().state_dict(*args, **kwargs)

    def set_eval(self):
        self._adopt_lists()
        for m in self.modules():
            m.training = False
            if isinstance(m, Module):
                m.is_train = False
        # Lazy device placement: the reference scripts never place the model (demo/image_classification/predict.py:16-20
        # goes load_weights -> set_eval -> predict); the engine only runs on the GPU, so a model still on the host moves
        # there the moment it is switched to inference.
        if torch.cuda.is_available():
            p = next(self.parameters(), None)
            if p is not None and not p.is_cuda:
                self.to("cuda")
        return self

    def set_train(self):
        self._adopt_lists()
        for m in self.modules():
            m.training = True
            if isinstance(m, Module):
                m.is_train = True
                m._engine_cache.clear()
        return self

    def eval(self):
        return self.set_eval()

    def train(self, mode=True):
        return self.set_train() if mode else self.set_eval()

    def _apply(self, fn, *a, **k):
        self._adopt_lists()
        self._invalidate()
        return super()._apply(fn, *a, **k)

    @property
    def all_weights(self):
        return [t for _, t in self.tlx_weights()]

    @property
    def trainable_weights(self):
        self._adopt_lists()
        return list(self.parameters())

    def _get_weights(self, var_name, shape, init=None, trainable=True, order=False):
        init = str_to_init(init) if isinstance(init, str) else (init or Constant(0.0))
        p = Parameter(data=init(shape=tuple(shape)))
        if trainable:
            self.register_parameter(var_name, p)
        else:
            self.register_buffer(var_name, p.data)
            p = getattr(self, var_name)
        return p

    def str_to_init(self, s):
        return str_to_init(s)

    def register_parameter(self, name=None, param=None):  # keyword form: vision_transformer.py:296
        return super().register_parameter(name, param)

    # Buffers that are functions of the architecture, not weights: never stored positionally, never required by name.
    _DERIVED = ("attn_mask", "relative_position_bias", "relative_position_index")

    def tlx_weights(self):
        """The model's weights in TensorLayerX's `all_weights` order [TLX-recalled: layers in construction order, each
        layer's variables in the order its build() creates them — conv/linear: weight then bias; BatchNorm: beta, gamma,
        moving_mean, moving_var; LayerNorm: gamma, beta — trainable and non-trainable interleaved].  This is the order of
        the positional `.npz` checkpoint (`save_weights('model.npz')`, demo/image_classification/train.py:55).
        Returns [(dotted name, tensor)]."""
        self._adopt_lists()
        out = []
        for prefix, m in self.named_modules():
            own = dict(m._parameters)
            own.update(m._buffers)
            order = getattr(m, "_TLX_ORDER", None) or list(own)
            for k in order:
                t = own.get(k)
                if t is None or k in Module._DERIVED:
                    continue
                out.append(((prefix + "." if prefix else "") + k, t))
        return out

    def save_weights(self, file_path, format=None):
        """format 'npz' (the default for a .npz path, as in TensorLayerX): ONE object array under the key `params`, the
        weights in tlx_weights() order, no names [TLX-recalled: tlx.files.save_npz].  format 'npz_dict': {dotted name:
        array}.  Conv filters are stored OIHW and Linear weights (in_features, out_features), as the torch backend of
        TensorLayerX keeps them [TLX-recalled]."""
        self._adopt_lists()
        fmt = format or "npz"
        if fmt == "npz":
            arrs = [t.detach().float().cpu().numpy() for _, t in self.tlx_weights()]
            params = np.empty(len(arrs), dtype=object)
            for i, a in enumerate(arrs):
                params[i] = a
            np.savez(file_path, params=params)
        elif fmt == "npz_dict":
            np.savez(file_path, **{k: v.detach().float().cpu().numpy() for k, v in self.state_dict().items()})
        else:
            raise NotImplementedError(f"save_weights: format {fmt!r} (npz and npz_dict are supported)")

    def load_weights(self, file_path, format=None, in_order=True, skip=False):
        """Loads either checkpoint form (detected from the file): the positional `params` list TensorLayerX's
        save_weights('x.npz') writes — assigned to tlx_weights() in order, every shape checked — or a name-keyed
        npz_dict (in_order=False semantics; `skip` tolerates missing names).  predict.py:19."""
        self._adopt_lists()
        # Name-keyed files hold plain arrays: opened WITHOUT pickle.  Only TensorLayerX's positional form — the single
        # object array `params` — needs the unpickler, so the file is re-opened with it for that form alone: loading such a
        # checkpoint runs pickle, i.e. it must come from a source you trust (as with TensorLayerX's own load_weights).
        data = np.load(file_path, allow_pickle=False)
        if list(data.files) == ["params"]:
            data.close()
            data = np.load(file_path, allow_pickle=True)
        with torch.no_grad():
            if "params" in data.files and len(data.files) == 1:
                params = list(data["params"])
                mine = self.tlx_weights()
                if len(params) != len(mine):
                    raise ValueError(f"load_weights: {file_path} holds {len(params)} arrays, the model has {len(mine)} weights")
                for (k, t), a in zip(mine, params):
                    a = torch.as_tensor(np.asarray(a))
                    if tuple(a.shape) != tuple(t.shape):
                        raise ValueError(f"load_weights: position of {k}: shape {tuple(a.shape)}, expected {tuple(t.shape)}")
                    t.copy_(a.to(t.dtype))
            else:
                sd = self.state_dict()
                missing = [k for k in sd if k not in data.files and k.rsplit(".", 1)[-1] not in Module._DERIVED]
                if missing and not skip:
                    raise KeyError(f"load_weights: {len(missing)} entries missing from {file_path}, e.g. {missing[:3]}")
                for k, t in sd.items():
                    if k in data.files:
                        a = torch.as_tensor(data[k])
                        if tuple(a.shape) != tuple(t.shape):
                            raise ValueError(f"load_weights: {k} has shape {tuple(a.shape)}, expected {tuple(t.shape)}")
                        t.copy_(a.to(t.dtype))
        self._invalidate()

    def load_dict(self, named, strict=True):
        """Assign {dotted name: numpy/torch array}; used by tests and benches with seeded recipes."""
        self._adopt_lists()
        sd = self.state_dict()
        derived = Module._DERIVED   # computed buffers
        unknown = [k for k in named if k not in sd]
        missing = [k for k in sd if k not in named and sd[k].is_floating_point()
                   and k.rsplit(".", 1)[-1] not in derived]
        if strict and (unknown or missing):
            raise KeyError(f"load_dict: unknown={unknown[:4]} missing={missing[:4]}")
        with torch.no_grad():
            for k, v in named.items():
                if k in sd:
                    a = torch.as_tensor(np.asarray(v)) if not isinstance(v, torch.Tensor) else v
                    if tuple(a.shape) != tuple(sd[k].shape):
                        raise ValueError(f"load_dict: {k} shape {tuple(a.shape)} != {tuple(sd[k].shape)}")
                    sd[k].copy_(a.to(sd[k].dtype))
        self._invalidate()

    def load_state_dict(self, *args, **kwargs):
        r = super().load_state_dict(*args, **kwargs)
        self._invalidate()
        return r

    def _require_eval(self):
        if self.is_train:
            raise NotImplementedError(
                f"{type(self).__name__}: training-mode forward is out of scope for the MI355X inference "
                "engine — call model.set_eval() first (tasks/image_classification.py:21 does).")

    def _invalidate(self):
        """Weights of this sub-tree were (re)loaded or moved: drop every derived tensor and advance the epoch a captured
        hipGraph (graph.GraphedForward) compares before each replay."""
        for m in self.modules():
            if isinstance(m, Module):
                m._engine_cache.clear()
                m._weights_epoch += 1

    def _stamp(self):
        """(storage address, in-place version) of every parameter / buffer this module owns: a derived tensor (packed
        filter, folded BatchNorm, bias table) is rebuilt when any of them was reassigned or written in place —
        load_state_dict, p.copy_(), p.data = ... — not only by this class's own loaders."""
        return tuple((t.data_ptr(), t._version) for t in list(self._parameters.values()) + list(self._buffers.values())
                     if t is not None)

    def _cached(self, key, build, deps=()):
        """Derived tensors of this layer, keyed by (key, precision) and validated against the parameter stamps of this
        module and of `deps` (other modules whose weights went into the value, e.g. the BatchNorm folded into a conv)."""
        k = (key, E.precision())
        stamp = self._stamp() + tuple(s_ for d in deps for s_ in d._stamp())
        hit = self._engine_cache.get(k)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        v = build()
        E.note_cache_build()       # built on the current stream: run_halves() redoes a forward during which this happened
        self._engine_cache[k] = (stamp, v)
        return v


class Identity(Module):
    def forward(self, x):
        return x


class Sequential(Module):
    """Accepts a list (resnet.py:247,284) or varargs (mobilenetv1.py:65,245)."""

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

    def __len__(self):
        return len(self._modules)

    def __iter__(self):
        return iter(self._modules.values())

    def __getitem__(self, i):
        return list(self._modules.values())[i]

    def append(self, layer):
        self.add_module(str(len(self._modules)), layer)

    def forward(self, x):
        for l in self._modules.values():
            x = l(x)
        return x


class ModuleList(torch.nn.ModuleList):
    def __init__(self, modules=None, name=None):
        super().__init__(modules)


# ---------------------------------------------------------------------------------------------
# layout helpers: logical NCHW (channels_first) <-> physical NHWC
# ---------------------------------------------------------------------------------------------
def as_nhwc(x, data_format="channels_first"):
    E.need_gpu(x, "input")
    dt = E.precision()
    if data_format == "channels_last":
        if x.dtype == dt and x.is_contiguous() and x.shape[-1] % E.vec(dt) == 0:
            return x
        return E.nchw_to_nhwc(x.permute(0, 3, 1, 2), dt)
    v = x.permute(0, 2, 3, 1)
    if x.dtype == dt and v.is_contiguous() and v.shape[-1] % E.vec(dt) == 0:
        return v
    return E.nchw_to_nhwc(x, dt)


def true_channels(x, data_format="channels_first"):
    return x.shape[-1] if data_format == "channels_last" else x.shape[1]


def from_nhwc(y, data_format="channels_first", channels=None):
    """Physical NHWC -> the caller's logical layout.  `channels`: the true channel count when as_nhwc() had to pad the
    channel axis to a whole number of 16-byte chunks (C = 3, 30, 291 ...): the padding is cropped off again."""
    if channels is not None and y.shape[-1] != channels:
        y = y[..., :channels]
    return y if data_format == "channels_last" else y.permute(0, 3, 1, 2)


def _tup2(v):
    return (v, v) if isinstance(v, int) else tuple(int(a) for a in v)


_ACTS = {}


class _Act(Module):
    ACT = E.ACT_NONE
    PARAM = 0.0

    def forward(self, x):
        return E.act_flat(x, self.ACT, self.PARAM)      # elementwise: any shape, any channel count, layout kept


class ReLU(_Act):  # resnet.py:50,138,212
    ACT = E.ACT_RELU


class ReLU6(_Act):
    ACT = E.ACT_RELU6


class Hardswish(_Act):
    ACT = E.ACT_HARDSWISH


class HardSigmoid(_Act):
    ACT = E.ACT_HARDSIGMOID


class Sigmoid(_Act):
    ACT = E.ACT_SIGMOID


class GELU(_Act):  # tlx.ops.GeLU, vision_transformer.py:70 — exact erf form [TLX-recalled]
    ACT = E.ACT_GELU


class LeakyReLU(_Act):  # darknet.py:50
    ACT = E.ACT_LEAKY

    def __init__(self, negative_slope=0.01, name=None):
        super().__init__(name=name)
        self.PARAM = float(negative_slope)


def str_to_act(act):
    """TensorLayerX layers take `act` as a callable or a name (resnext.py:46-52 BatchNorm(act='relu'))."""
    if act is None or not isinstance(act, str):
        return act
    table = {"relu": ReLU, "relu6": ReLU6, "sigmoid": Sigmoid, "gelu": GELU, "hardswish": Hardswish,
             "hard_sigmoid": HardSigmoid, "leaky_relu": LeakyReLU}
    if act.lower() not in table:
        raise NotImplementedError(f"activation {act!r}")
    return table[act.lower()]()


class Softmax(Module):
    def __init__(self, axis=-1, name=None):
        super().__init__(name=name)
        self.axis = axis

    def forward(self, x):
        return E.softmax(x, self.axis)


class Dropout(Module):
    """Identity in eval mode (vision_transformer.py:80,109,111)."""

    def __init__(self, p=0.5, seed=0, name=None):
        super().__init__(name=name)
        self.p = p

    def forward(self, x):
        if self.is_train and self.p > 0:
            self._require_eval()
        return x


class Flatten(Module):  # tlx.FlattenReshape, resnet.py:232
    def forward(self, x):
        return x.reshape(x.shape[0], -1)


# ---------------------------------------------------------------------------------------------
# GroupConv2d / Conv2d
# ---------------------------------------------------------------------------------------------
class GroupConv2d(Module):
    """nn.GroupConv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, n_group,
    b_init, W_init, data_format, act, name) — call sites resnet.py:37,99,111,126,199,248;
    vision_transformer.py:197; mobilenetv1.py:47; darknet.py:38; yolov3.py:313."""

    def __init__(self, out_channels=32, kernel_size=(1, 1), stride=(1, 1), n_group=1, act=None, padding="SAME",
                 data_format="channels_last", dilation=(1, 1), W_init="truncated_normal", b_init="constant",
                 in_channels=None, name=None):
        super().__init__(name=name)
        if in_channels is None:
            raise ValueError("GroupConv2d: in_channels must be given (deferred build is not supported)")
        self.in_channels, self.out_channels = int(in_channels), int(out_channels)
        self.kernel_size, self.stride, self.dilation = _tup2(kernel_size), _tup2(stride), _tup2(dilation)
        self.n_group = int(n_group)
        if self.in_channels % self.n_group or self.out_channels % self.n_group:
            raise ValueError("The number of input/output channels must be divisible by n_group")
        self.same = False
        if isinstance(padding, str):
            p = padding.upper()
            if p == "VALID":
                self.padding = (0, 0)
            elif p == "SAME":
                # TensorFlow's rule [TLX-recalled: conv2d_same_padding of the torch backend]: total padding
                # max(0, (ceil(in / s) - 1) * s + d * (k - 1) + 1 - in), the odd unit at the bottom / right.  For odd
                # kernels at stride 1 that is the symmetric value stored here; other cases are resolved per input
                # size in _same() (efficientnet.py:92-125: 'SAME' at stride 2).
                self.same = True
                self.padding = tuple(d * (k - 1) // 2 for k, d in zip(self.kernel_size, self.dilation))
            else:
                raise ValueError(f"unsupported padding {padding!r}")
        else:
            self.padding = _tup2(padding)
        self.data_format = data_format
        self.act = act
        w_init = str_to_init(W_init) if isinstance(W_init, str) else W_init
        shape = (self.out_channels, self.in_channels // self.n_group) + self.kernel_size
        self.filters = Parameter(data=w_init(shape=shape))
        if b_init is None or b_init is False or (isinstance(b_init, (tuple, list)) and len(b_init) == 0):
            self.biases = None
        else:
            bi = str_to_init(b_init) if isinstance(b_init, str) else (b_init if callable(b_init) else Constant(0.0))
            self.biases = Parameter(data=bi(shape=(self.out_channels,)))

    def _same(self, H, W):
        """padding='SAME' for this input size -> (leading padding, output extent or None when the symmetric form is exact)."""
        lead, out, odd = [], [], False
        for i, k, s_, d in zip((H, W), self.kernel_size, self.stride, self.dilation):
            o = -(-i // s_)
            total = max(0, (o - 1) * s_ + d * (k - 1) + 1 - i)
            lead.append(total // 2)
            out.append(o)
            odd = odd or (total % 2 == 1)
        return tuple(lead), (tuple(out) if odd else None)

    # fused entry point used by the model graphs: conv (+bn) (+act) (+residual) in one launch
    def run_nhwc(self, x, bn=None, act=E.ACT_NONE, act_param=0.0, res=None, res_after_act=False, **kw):
        self._require_eval()
        dt = E.precision()
        padding = self.padding
        if self.same:
            padding, out_hw = self._same(x.shape[1], x.shape[2])
            if out_hw is not None:
                kw = dict(kw, out_hw=out_hw)
                if self.n_group == 1 or self.n_group != self.in_channels:
                    kw["overhang"] = True
        if self.n_group == 1:
            pk = self._cached("pk", lambda: E.PackedFilter(self.filters, dt))
        elif self.n_group == self.in_channels == self.out_channels:
            pk = self._cached("dw", lambda: self.filters.detach()[:, 0].permute(1, 2, 0).contiguous().to(dt))
        else:   # 1 < n_group < C: ResNeXt cardinality, resnext.py:83-91
            pk = self._cached("gpk", lambda: E.PackedGroupFilter(self.filters, self.n_group, dt))
        if bn is not None:
            scale, shift = self._cached(("bn", id(bn)), lambda: bn.folded(self.biases), deps=(bn,))
        else:
            scale, shift = None, (self._cached("bias", lambda: E._f32(self.biases)) if self.biases is not None else None)
        if self.n_group == 1:
            return E.conv2d(x, pk, self.stride, padding, self.dilation, scale, shift, res, act, act_param,
                            res_after_act, **kw)
        if isinstance(pk, E.PackedGroupFilter):
            if kw.get("out_hw") is not None:
                raise NotImplementedError("grouped conv with one-sided 'SAME' padding")
            return E.group_conv2d(x, pk, self.stride, padding, self.dilation, scale, shift, res, act, act_param,
                                  res_after_act)
        if res is not None:
            raise NotImplementedError("depthwise conv with fused residual")
        return E.dwconv2d(x, pk, self.stride, padding, self.dilation, scale, shift, act, act_param, out_hw=kw.get("out_hw"))

    def run_stem(self, x_nchw, b, bn=None, act=E.ACT_NONE, act_param=0.0, maxpool=None, **kw):
        """Few-channel first conv (RGB stem / patch embedding) on a b x b space-to-depth input: the 3-channel
        image would leave 5 of every 8 fp16 K-lanes zero; folding b x b pixels into channels makes K dense
        (7x7/2 stem: 448 -> 256; 16x16/16 patch embed: 2048 -> 768).  Same arithmetic, re-indexed once.
        maxpool: the nn.MaxPool2d that follows (resnet.py:290) — taken in the conv's epilogue when it is the 3/2/1 pool
        and the library has the fused kernel for this geometry, otherwise run as its own launch."""
        self._require_eval()
        E.need_gpu(x_nchw, "input")
        if self.data_format != "channels_first" or self.n_group != 1 or self.dilation != (1, 1):
            raise NotImplementedError("run_stem: channels_first, dense, undilated convs only")
        sh, sw = self.stride
        N, Cc, H, W = x_nchw.shape
        pad, same_hw = self.padding, None
        if self.same:       # 'SAME' (efficientnet.py:354-363): leading padding for this size; an odd total runs past the end
            pad, same_hw = self._same(H, W)
        if sh % b or sw % b or sh != sw or pad[0] != pad[1]:
            raise NotImplementedError("run_stem: stride must be a multiple of the fold")
        dt = E.precision()
        Ho = (H + 2 * pad[0] - self.kernel_size[0]) // sh + 1
        Wo = (W + 2 * pad[1] - self.kernel_size[1]) // sw + 1
        if same_hw is not None:
            Ho, Wo = same_hw
            kw = dict(kw, overhang=True)

        def build():
            w2, pad2 = E.s2d_filter(self.filters, b, pad[0])
            return E.PackedFilter(w2, dt), pad2
        pk, pad2 = self._cached(("s2d", b, pad[0]), build)
        if bn is not None:
            scale, shift = self._cached(("bn", id(bn)), lambda: bn.folded(self.biases), deps=(bn,))
        else:
            scale, shift = None, (self._cached("bias", lambda: E._f32(self.biases)) if self.biases is not None else None)
        v = E.nchw_to_nhwc_s2d(x_nchw, b, dt)
        if maxpool is not None and (maxpool.kernel_size, maxpool.stride, maxpool.padding) == ((3, 3), (2, 2), (1, 1)) and not kw:
            y = E.conv2d(v, pk, (sh // b, sw // b), pad2, 1, scale, shift, act=act, act_param=act_param, out_hw=(Ho, Wo),
                         maxpool3s2=True)
            if y is not None:
                return y
        y = E.conv2d(v, pk, (sh // b, sw // b), pad2, 1, scale, shift, act=act, act_param=act_param,
                     out_hw=(Ho, Wo), **kw)
        return maxpool.run_nhwc(y) if maxpool is not None else y

    def forward(self, x):
        y = self.run_nhwc(as_nhwc(x, self.data_format))
        y = from_nhwc(y, self.data_format)
        return self.act(y) if self.act is not None else y


class Conv2d(GroupConv2d):
    def __init__(self, out_channels=32, kernel_size=(3, 3), stride=(1, 1), act=None, padding="SAME",
                 data_format="channels_last", dilation=(1, 1), W_init="truncated_normal", b_init="constant",
                 in_channels=None, name=None):
        super().__init__(out_channels=out_channels, kernel_size=kernel_size, stride=stride, n_group=1, act=act,
                         padding=padding, data_format=data_format, dilation=dilation, W_init=W_init, b_init=b_init,
                         in_channels=in_channels, name=name)


# ---------------------------------------------------------------------------------------------
# BatchNorm2d
# ---------------------------------------------------------------------------------------------
class BatchNorm2d(Module):
    """nn.BatchNorm2d(num_features=, data_format=, epsilon=, momentum=) — resnet.py:46,107,122,134,208,257;
    mobilenetv1.py:61; darknet.py:48; mobilenetv3.py:148 (eps 1e-3)."""

    def __init__(self, momentum=0.9, epsilon=1e-5, act=None, is_train=True, beta_init="zeros", gamma_init="ones",
                 moving_mean_init="zeros", moving_var_init="ones", num_features=None, data_format="channels_last",
                 name=None):
        super().__init__(name=name)
        if num_features is None:
            raise ValueError("BatchNorm2d: num_features must be given")
        self.num_features, self.epsilon, self.momentum = int(num_features), float(epsilon), momentum
        self.data_format, self.act = data_format, str_to_act(act)
        self._TLX_ORDER = ("beta", "gamma", "moving_mean", "moving_var")
        n = (self.num_features,)
        self.gamma = Parameter(data=str_to_init(gamma_init)(shape=n))
        self.beta = Parameter(data=str_to_init(beta_init)(shape=n))
        self.register_buffer("moving_mean", str_to_init(moving_mean_init)(shape=n))
        self.register_buffer("moving_var", str_to_init(moving_var_init)(shape=n))

    def folded(self, conv_bias=None):
        return E.fold_bn(self.gamma, self.beta, self.moving_mean, self.moving_var, self.epsilon, conv_bias)

    def forward(self, x):
        self._require_eval()
        v = as_nhwc(x, self.data_format)
        cpad = v.shape[-1]

        def padded():       # (scale, shift) over the padded channel axis: the kernel reads v.shape[-1] of each
            sc, sh = self.folded()
            if cpad == self.num_features:
                return sc, sh
            z = torch.zeros(cpad - self.num_features, dtype=sc.dtype, device=sc.device)
            return torch.cat([sc, z]).contiguous(), torch.cat([sh, z]).contiguous()
        scale, shift = self._cached(("fold", cpad), padded)
        y = from_nhwc(E.affine_act(v, scale, shift), self.data_format, self.num_features)
        return self.act(y) if self.act is not None else y


BatchNorm = BatchNorm2d


class LayerNorm(Module):
    """nn.LayerNorm(normalized_shape, epsilon=) — vision_transformer.py:144,159,283; swin :258,279."""

    def __init__(self, normalized_shape, epsilon=1e-5, gamma_init="ones", beta_init="zeros", act=None, name=None):
        super().__init__(name=name)
        if isinstance(normalized_shape, int):
            normalized_shape = (normalized_shape,)
        self.normalized_shape, self.epsilon = tuple(normalized_shape), float(epsilon)
        if len(self.normalized_shape) != 1:
            raise NotImplementedError("LayerNorm over more than the last axis")
        self.gamma = Parameter(data=str_to_init(gamma_init)(shape=self.normalized_shape))
        self.beta = Parameter(data=str_to_init(beta_init)(shape=self.normalized_shape))

    def forward(self, x):
        E.need_gpu(x, "input")
        if x.dtype != E.precision():
            x = x.to(E.precision())
        return E.layernorm(x, self.gamma.detach(), self.beta.detach(), self.epsilon)


class Linear(Module):
    """nn.Linear(in_features=, out_features=, b_init=) — resnet.py:234; vision_transformer.py:77,79,105-110,
    284; `b_init=False/None` => no bias (:107)."""

    def __init__(self, out_features=None, act=None, W_init="truncated_normal", b_init="constant", in_features=None,
                 name=None):
        super().__init__(name=name)
        if in_features is None or out_features is None:
            raise ValueError("Linear: in_features and out_features must be given")
        self.in_features, self.out_features, self.act = int(in_features), int(out_features), act
        w_init = str_to_init(W_init) if isinstance(W_init, str) else W_init
        self.weights = Parameter(data=w_init(shape=(self.in_features, self.out_features)))
        if b_init is None or b_init is False or (isinstance(b_init, (tuple, list)) and len(b_init) == 0):
            self.biases = None
        else:
            bi = str_to_init(b_init) if isinstance(b_init, str) else (b_init if callable(b_init) else Constant(0.0))
            self.biases = Parameter(data=bi(shape=(self.out_features,)))

    def run(self, x, act=E.ACT_NONE, res=None, out=None):
        dt = E.precision()
        pk = self._cached("pk", lambda: E.PackedFilter(self.weights.detach().t().contiguous(), dt))
        b = self._cached("bias", lambda: E._f32(self.biases)) if self.biases is not None else None
        if x.dtype != dt:
            x = x.to(dt)
        return E.linear(x, pk, b, res, act, out)

    # LayerNorm folded around the Linear layers of a transformer block (engine.linear_stats / linear_ln, fp16, round 5)
    def run_stats(self, x, res=None, out=None):
        """run() of a Linear that writes the residual stream, + the row statistics of its output -> (y, partials)."""
        dt = E.precision()
        pk = self._cached("pk", lambda: E.PackedFilter(self.weights.detach().t().contiguous(), dt))
        b = self._cached("bias", lambda: E._f32(self.biases)) if self.biases is not None else None
        return E.linear_stats(x, pk, b, res, out)

    def run_ln(self, x, norm, part, act=E.ACT_NONE):
        """act(self(norm(x))) on the raw rows x, `norm` (an nn.LayerNorm) folded in: part = the row statistics of x (run_stats of whoever wrote x)."""
        dt = E.precision()
        prep = self._cached(("ln_fold", id(norm)), lambda: E.LinearLN(self.weights.detach().t().contiguous(), self.biases.detach() if self.biases is not None else None,
                                                                      norm.gamma, norm.beta, dt), deps=(norm,))
        return E.linear_ln(x, prep, part, norm.epsilon, act)

    def forward(self, x):
        y = self.run(x)
        return self.act(y) if self.act is not None else y


class MaxPool2d(Module):
    """nn.MaxPool2d(kernel_size, stride, padding, data_format) — resnet.py:213-218."""

    def __init__(self, kernel_size, stride=None, padding="SAME", return_mask=False, data_format="channels_last",
                 name=None):
        super().__init__(name=name)
        self.kernel_size = _tup2(kernel_size)
        self.stride = _tup2(stride if stride is not None else kernel_size)
        if isinstance(padding, str):
            self.padding = (0, 0) if padding.upper() == "VALID" else tuple((k - 1) // 2 for k in self.kernel_size)
        else:
            self.padding = _tup2(padding)
        self.data_format = data_format

    def run_nhwc(self, x):
        return E.maxpool2d(x, self.kernel_size, self.stride, self.padding)

    def forward(self, x):
        return from_nhwc(self.run_nhwc(as_nhwc(x, self.data_format)), self.data_format, true_channels(x, self.data_format))


class AvgPool2d(Module):
    """nn.AvgPool2d(kernel_size, stride, padding, data_format) — resnest.py:212-218, 250-256, 271-286; zero padding counts
    in the divisor [TLX-recalled: the torch backend forwards to F.avg_pool2d's default]."""

    def __init__(self, kernel_size, stride=None, padding="SAME", ceil_mode=False, data_format="channels_last", name=None):
        super().__init__(name=name)
        self.kernel_size = _tup2(kernel_size)
        self.stride = _tup2(stride if stride is not None else kernel_size)
        if isinstance(padding, str):
            self.padding = (0, 0) if padding.upper() == "VALID" else tuple((k - 1) // 2 for k in self.kernel_size)
        else:
            self.padding = _tup2(padding)
        self.data_format = data_format

    def run_nhwc(self, v):
        if self.kernel_size == (1, 1) and self.stride == (1, 1):
            return v
        return E.avgpool2d(v, self.kernel_size, self.stride, self.padding)

    def forward(self, x):
        return from_nhwc(self.run_nhwc(as_nhwc(x, self.data_format)), self.data_format, true_channels(x, self.data_format))


class AdaptiveAvgPool2d(Module):
    """nn.AdaptiveAvgPool2d(output_size, data_format): (1,1) on the hot path (resnet.py:228-231, mobilenetv1.py:246),
    (7,7) in VGG (vgg.py:36-39; the identity at 224 x 224)."""

    def __init__(self, output_size, data_format="channels_last", name=None):
        super().__init__(name=name)
        self.output_size = _tup2(output_size)
        self.data_format = data_format

    def run_nhwc(self, v):
        if self.output_size == (1, 1):
            N = v.shape[0]
            return E.global_avgpool(v).view(N, 1, 1, -1)
        return E.adaptive_avgpool2d(v, self.output_size)

    def forward(self, x):
        return from_nhwc(self.run_nhwc(as_nhwc(x, self.data_format)), self.data_format, true_channels(x, self.data_format))


class AdaptiveAvgPool1d(Module):
    """swin_transformer.py:593,609: mean over tokens of a (B, C, L) tensor."""

    def __init__(self, output_size, data_format="channels_first", name=None):
        super().__init__(name=name)
        if output_size != 1:
            raise NotImplementedError("AdaptiveAvgPool1d: only output_size 1")
        self.data_format = data_format

    def forward(self, x):
        E.need_gpu(x, "input")
        if self.data_format == "channels_first":  # (B, C, L)
            y = E.global_avgpool(x.transpose(1, 2).contiguous())
            return y.unsqueeze(-1)
        return E.global_avgpool(x.contiguous()).unsqueeze(1)


class UpSampling2d(Module):
    def __init__(self, scale=2, method="nearest", data_format="channels_first", name=None):
        super().__init__(name=name)
        if _tup2(scale) != (2, 2) or method != "nearest":
            raise NotImplementedError("UpSampling2d: only nearest x2 (yolov3.py:250)")
        self.data_format = data_format

    def forward(self, x):
        v = as_nhwc(x, self.data_format)
        N, H, W, Cc = v.shape
        out = torch.empty((N, 2 * H, 2 * W, Cc), dtype=v.dtype, device=v.device)
        return from_nhwc(E.upsample2x_into(v, out, 0), self.data_format, true_channels(x, self.data_format))


class MultiheadAttention(Module):
    """tlx.nn.MultiheadAttention(embed_dim, num_heads, dropout, kdim, vdim, bias, batch_first, need_weights) — named by
    BASELINE.json's north_star; the reference's own consumer of the same computation is DETR's MultiHeadAttention
    (tlxcv/models/detection/detr.py:965-1062), whose arithmetic this follows: q / k / v projections with bias, q scaled by
    head_dim^-0.5 before q k^T, additive attn_mask, softmax, @ v, output projection, weights averaged over the heads.
    Parameter names and shapes [TLX-recalled]: q_weight (E, E), k_weight (E, kdim), v_weight (E, vdim), out_weight (E, E)
    stored (out, in) as torch's functional form takes them, q_bias / k_bias / v_bias / out_bias.
    forward(q, k=None, v=None, attn_mask=None, key_padding_mask=None) -> (attn_output, attn_weights or None); inputs are
    (L, B, E) unless batch_first.  The three projections and the output projection are GEMM launches, the core is
    tlxmi_mha — or, for self attention of <= 256 tokens without mask / weights in fp16, the fused tlxmi_attention."""

    def __init__(self, embed_dim, num_heads, dropout=0.0, kdim=None, vdim=None, bias=True, batch_first=False,
                 need_weights=True, name=None):
        super().__init__(name=name)
        self.embed_dim, self.num_heads = int(embed_dim), int(num_heads)
        self.kdim = int(kdim) if kdim is not None else self.embed_dim
        self.vdim = int(vdim) if vdim is not None else self.embed_dim
        if self.embed_dim % self.num_heads:
            raise ValueError("embed_dim must be divisible by num_heads")
        self.head_dim = self.embed_dim // self.num_heads
        self.dropout, self.batch_first, self.need_weights, self.bias = dropout, batch_first, need_weights, bias
        init = xavier_uniform()
        E_ = self.embed_dim
        self.q_weight = Parameter(data=init(shape=(E_, E_)))
        self.k_weight = Parameter(data=init(shape=(E_, self.kdim)))
        self.v_weight = Parameter(data=init(shape=(E_, self.vdim)))
        self.out_weight = Parameter(data=init(shape=(E_, E_)))
        if bias:
            for n in ("q_bias", "k_bias", "v_bias", "out_bias"):
                setattr(self, n, Parameter(data=Constant(0.0)(shape=(E_,))))
        else:
            self.q_bias = self.k_bias = self.v_bias = self.out_bias = None

    def _proj(self, x, wname, bname, res=None):
        dt = E.precision()
        w = getattr(self, wname)
        pk = self._cached(("pk", wname), lambda: E.PackedFilter(w.detach().contiguous(), dt))
        b = getattr(self, bname)
        bb = self._cached(("b", bname), lambda: E._f32(b)) if b is not None else None
        return E.linear(x if x.dtype == dt else x.to(dt), pk, bb)

    def forward(self, q, k=None, v=None, attn_mask=None, key_padding_mask=None):
        self._require_eval() if self.dropout else None
        E.need_gpu(q, "query")
        k = q if k is None else k
        v = k if v is None else v
        if key_padding_mask is not None:
            raise NotImplementedError("MultiheadAttention: key_padding_mask (fold it into attn_mask as -inf columns)")
        dt = E.precision()
        hd, H = self.head_dim, self.num_heads
        self_attn = (k is q) and (v is q)
        L = q.shape[1] if self.batch_first else q.shape[0]
        if self_attn and attn_mask is None and not self.need_weights and dt == torch.float16 and L <= 256 and hd in (32, 64, 96):
            # fused path: one packed qkv GEMM on batch-first rows + the MFMA attention kernel
            xb = (q if self.batch_first else q.transpose(0, 1)).to(dt).contiguous()          # (B, L, E)
            pk = self._cached("pk_qkv", lambda: E.PackedFilter(
                torch.cat([self.q_weight, self.k_weight, self.v_weight], 0).detach().contiguous(), dt))
            bb = self._cached("b_qkv", lambda: E._f32(torch.cat([self.q_bias, self.k_bias, self.v_bias]))) if self.bias else None
            a = E.attention(E.linear(xb, pk, bb), H, hd ** -0.5)
            o = self._proj(a, "out_weight", "out_bias")
            return (o if self.batch_first else o.transpose(0, 1)), None
        Q = self._proj(q, "q_weight", "q_bias")
        K = self._proj(k, "k_weight", "k_bias")
        V = self._proj(v, "v_weight", "v_bias")
        a, w = E.mha(Q, K, V, H, hd ** -0.5, attn_mask, self.need_weights, self.batch_first)
        return self._proj(a, "out_weight", "out_bias"), w
