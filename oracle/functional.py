"""fp32 torch-CPU restatements of the reference forward graphs (TEST INFRASTRUCTURE — see
oracle/__init__.py).  Parameters are a flat {dotted attribute path: torch.Tensor} dictionary with
the reference's module tree names (e.g. 'layer1.0.conv1.filters'), as produced by
tlxcv_amd.seeded.fill.  Citations are into /root/reference/tlxcv/.

TensorLayerX layer semantics assumed (restated from its documentation; not verifiable here):
conv = cross-correlation with symmetric zero padding; BatchNorm eval = (x-mean)/sqrt(var+eps)*g+b,
eps 1e-5; MaxPool pads -inf; LayerNorm biased variance; GELU exact erf; Linear y = x @ W(in,out) + b.
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _t(p, name):
    v = p[name]
    return v if isinstance(v, torch.Tensor) else torch.as_tensor(v)


def conv(p, name, x, stride=1, padding=0, dilation=1, groups=1):
    """nn.GroupConv2d forward (bias only if the layer has one)."""
    b = _t(p, name + ".biases") if (name + ".biases") in p else None
    return F.conv2d(x, _t(p, name + ".filters"), b, stride, padding, dilation, groups)


def bn(p, name, x, eps=BN_EPS):
    """nn.BatchNorm2d in eval mode."""
    return F.batch_norm(x, _t(p, name + ".moving_mean"), _t(p, name + ".moving_var"), _t(p, name + ".gamma"),
                        _t(p, name + ".beta"), False, 0.0, eps)


def linear(p, name, x):
    y = torch.matmul(x, _t(p, name + ".weights"))
    if (name + ".biases") in p:
        y = y + _t(p, name + ".biases")
    return y


def layernorm(p, name, x, eps):
    g = _t(p, name + ".gamma")
    return F.layer_norm(x, g.shape, g, _t(p, name + ".beta"), eps)


# ---------------------------------------------------------------------------------------------
# ResNet — models/classification/resnet.py
# ---------------------------------------------------------------------------------------------
RESNET_CFG = {18: ("basic", [2, 2, 2, 2]), 34: ("basic", [3, 4, 6, 3]), 50: ("bottleneck", [3, 4, 6, 3]),
              101: ("bottleneck", [3, 4, 23, 3]), 152: ("bottleneck", [3, 8, 36, 3])}  # resnet.py:184-190


def _bottleneck(p, pre, x, stride, has_down):
    """BottleneckBlock.forward, resnet.py:142-156 (stride on the 3x3, :111-121)."""
    identity = x
    out = F.relu(bn(p, pre + ".bn1", conv(p, pre + ".conv1", x)))                           # :144-146
    out = F.relu(bn(p, pre + ".bn2", conv(p, pre + ".conv2", out, stride, 1)))              # :147-149
    out = bn(p, pre + ".bn3", conv(p, pre + ".conv3", out))                                  # :150-151
    if has_down:                                                                             # :152-153
        identity = bn(p, pre + ".downsample.1", conv(p, pre + ".downsample.0", x, stride))  # :246-261
    out = out + identity                                                                     # :154
    return F.relu(out)                                                                       # :155


def _basic(p, pre, x, stride, has_down):
    """BasicBlock.forward, resnet.py:66-77."""
    identity = x
    out = F.relu(bn(p, pre + ".bn1", conv(p, pre + ".conv1", x, stride, 1)))
    out = bn(p, pre + ".bn2", conv(p, pre + ".conv2", out, 1, 1))
    if has_down:
        identity = bn(p, pre + ".downsample.1", conv(p, pre + ".downsample.0", x, stride))
    return F.relu(out + identity)


def resnet(p, x, depth=50, num_classes=1000, with_pool=True):
    """ResNet.forward, resnet.py:286-300; _make_layer :239-284."""
    kind, layers = RESNET_CFG[depth]
    block = _bottleneck if kind == "bottleneck" else _basic
    x = F.relu(bn(p, "bn1", conv(p, "conv1", x, 2, 3)))        # :287-289 (7x7/2 pad 3, :199-207)
    x = F.max_pool2d(x, 3, 2, 1)                                # :290 (:213-218)
    for li, (n, stride) in enumerate(zip(layers, (1, 2, 2, 2)), start=1):
        for bi in range(n):
            pre = f"layer{li}.{bi}"
            x = block(p, pre, x, stride if bi == 0 else 1, (pre + ".downsample.0.filters") in p)
    if with_pool:
        x = F.adaptive_avg_pool2d(x, (1, 1))                    # :295-296
    if num_classes > 0:
        x = x.reshape(x.shape[0], -1)                           # :298 tlx.FlattenReshape
        x = linear(p, "fc", x)                                  # :299
    return x


def predict(logits):
    """ImageClassification.predict, tasks/image_classification.py:20-23."""
    return torch.argmax(logits, dim=-1)


# ---------------------------------------------------------------------------------------------
# Single-op references used by the per-kernel parity tests (same math as the layers above)
# ---------------------------------------------------------------------------------------------
ACTS = {
    0: lambda x, a: x,
    1: lambda x, a: F.relu(x),
    2: lambda x, a: F.relu6(x),
    3: lambda x, a: F.leaky_relu(x, a),
    4: lambda x, a: F.hardswish(x),
    5: lambda x, a: F.hardsigmoid(x),
    6: lambda x, a: F.gelu(x, approximate="none"),
    7: lambda x, a: torch.sigmoid(x),
    8: lambda x, a: F.silu(x),
}


def conv_bn_act(x_nchw, w_oihw, scale=None, shift=None, res=None, act=0, act_param=0.0, stride=1, padding=0,
                dilation=1, groups=1, res_after_act=False):
    y = F.conv2d(x_nchw, w_oihw, None, stride, padding, dilation, groups)
    if scale is not None:
        y = y * scale.view(1, -1, 1, 1)
    if shift is not None:
        y = y + shift.view(1, -1, 1, 1)
    if res is not None and not res_after_act:
        y = y + res
    y = ACTS[act](y, act_param)
    if res is not None and res_after_act:
        y = y + res
    return y


def fold_bn(gamma, beta, mean, var, eps, conv_bias=None):
    scale = gamma / torch.sqrt(var + eps)
    shift = beta - mean * scale
    if conv_bias is not None:
        shift = shift + conv_bias * scale
    return scale, shift
