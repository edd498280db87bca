"""fp32 torch-CPU restatements of the reference forward graphs (TEST INFRASTRUCTURE — see
oracle/__init__.py).  Parameters are a flat {dotted attribute path: torch.Tensor} dictionary with
the reference's module tree names (e.g. 'layer1.0.conv1.filters'), as produced by
tlxcv_amd.seeded.fill.  Citations are into /root/reference/tlxcv/.

TensorLayerX layer semantics assumed (restated from its documentation; not verifiable here):
conv = cross-correlation with symmetric zero padding; BatchNorm eval = (x-mean)/sqrt(var+eps)*g+b,
eps 1e-5; MaxPool pads -inf; LayerNorm biased variance; GELU exact erf; Linear y = x @ W(in,out) + b.
"""
import math

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


# ---------------------------------------------------------------------------------------------
# VGG — models/classification/vgg.py;  AlexNet — models/classification/alexnet.py
# ---------------------------------------------------------------------------------------------
VGG_CFG = {  # vgg.py:93-98
    "A": [64, "M", 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M"],
    "B": [64, 64, "M", 128, 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M"],
    "D": [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"],
    "E": [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"],
}
VGG_ARCH = {"vgg11": "A", "vgg13": "B", "vgg16": "D", "vgg19": "E"}


def vgg(p, x, arch="vgg16", batch_norm=False, num_classes=1000, with_pool=True):
    """VGG.forward vgg.py:52-59 over make_layers :61-90: Sequential index i counts every layer
    (conv, [bn], relu, pool), which is how the parameters are named (`features.<i>.filters`)."""
    i = 0
    for v in VGG_CFG[VGG_ARCH[arch]]:
        if v == "M":
            x = F.max_pool2d(x, 2, 2)                                         # :66-72 (padding 'SAME' of a 2x2/2 pool = 0)
            i += 1
        else:
            x = conv(p, f"features.{i}", x, 1, 1)                             # :74-80 3x3, padding='SAME', bias
            i += 1
            if batch_norm:
                x = bn(p, f"features.{i}", x)                                 # :82-86
                i += 1
            x = F.relu(x)
            i += 1
    if with_pool:
        x = F.adaptive_avg_pool2d(x, (7, 7))                                  # :54-55
    if num_classes > 0:
        x = x.reshape(x.shape[0], -1)                                         # :56 FlattenReshape (C, H, W order)
        x = F.relu(linear(p, "classifier.0", x))                              # :57, :42-50 (Dropout = identity in eval)
        x = F.relu(linear(p, "classifier.3", x))
        x = linear(p, "classifier.6", x)
    return x


def alexnet(p, x, num_classes=1000):
    """AlexNet.forward alexnet.py:152-169; ConvPoolLayer.forward :44-49."""
    x = F.max_pool2d(F.relu(conv(p, "_conv1._conv", x, 4, 2)), 3, 2)          # :153  11x11/4 pad 2, pool 3/2/0
    x = F.max_pool2d(F.relu(conv(p, "_conv2._conv", x, 1, 2)), 3, 2)          # :154  5x5 pad 2
    x = F.relu(conv(p, "_conv3", x, 1, 1))                                    # :155-156
    x = F.relu(conv(p, "_conv4", x, 1, 1))                                    # :157-158
    x = F.max_pool2d(F.relu(conv(p, "_conv5._conv", x, 1, 1)), 3, 2)          # :159
    if num_classes > 0:
        x = x.flatten(1)                                                      # :161
        x = F.relu(linear(p, "_fc6", x))                                      # :162-164
        x = F.relu(linear(p, "_fc7", x))                                      # :165-167
        x = linear(p, "_fc8", x)                                              # :168
    return x


# ---------------------------------------------------------------------------------------------
# ResNeXt — models/classification/resnext.py
# ---------------------------------------------------------------------------------------------
RESNEXT_DEPTH = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}   # resnext.py:141-146


def _convbn(p, pre, x, stride=1, groups=1, relu=False):
    """ConvBNLayer.forward resnext.py:54-57: GroupConv2d (no bias, padding (k-1)//2 :35) + BatchNorm(act) :46-52."""
    k = _t(p, pre + "._conv.filters").shape[-1]
    y = bn(p, pre + ".batch_norm", conv(p, pre + "._conv", x, stride, (k - 1) // 2, 1, groups))
    return F.relu(y) if relu else y


def resnext(p, x, layers=50, cardinality=32):
    """ResNeXt.forward resnext.py:205-213; BottleneckBlock.forward :109-119."""
    x = _convbn(p, "conv", x, 2, 1, True)                                     # :206  7x7/2 pad 3 + bn + relu
    x = F.max_pool2d(x, 3, 2, 1)                                              # :207
    for block, n in enumerate(RESNEXT_DEPTH[layers]):
        for i in range(n):
            pre = f"bb_{block}_{i}"
            stride = 2 if i == 0 and block != 0 else 1                        # :181
            y = _convbn(p, pre + ".conv0", x, 1, 1, True)                     # :110  1x1
            y = _convbn(p, pre + ".conv1", y, stride, cardinality, True)      # :111  3x3, groups = cardinality
            y = _convbn(p, pre + ".conv2", y)                                 # :112  1x1, no activation
            short = x if i > 0 else _convbn(p, pre + ".short", x, stride)     # :113-116 (first block of a stage projects)
            x = F.relu(short + y)                                             # :117-118
    x = F.adaptive_avg_pool2d(x, 1).reshape(x.shape[0], -1)                   # :210-211
    return linear(p, "out", x)                                                # :212


# ---------------------------------------------------------------------------------------------
# EfficientNet-B0..B7 — models/classification/efficientnet.py
# ---------------------------------------------------------------------------------------------
EFFNET_ARCH = {   # efficientnet.py:465-547: (width_mult, depth_mult, BatchNorm epsilon)
    "efficientnet_b0": (1.0, 1.0, 1e-5), "efficientnet_b1": (1.0, 1.1, 1e-5), "efficientnet_b2": (1.1, 1.2, 1e-5),
    "efficientnet_b3": (1.2, 1.4, 1e-5), "efficientnet_b4": (1.4, 1.8, 1e-5), "efficientnet_b5": (1.6, 2.2, 1e-3),
    "efficientnet_b6": (1.8, 2.6, 1e-3), "efficientnet_b7": (2.0, 3.1, 1e-3),
}
EFFNET_STAGES = [  # efficientnet.py:446-454: expand ratio, kernel, stride, in, out, layers
    (1, 3, 1, 32, 16, 1), (6, 3, 2, 16, 24, 2), (6, 5, 2, 24, 40, 2), (6, 3, 2, 40, 80, 3),
    (6, 5, 1, 80, 112, 3), (6, 5, 2, 112, 192, 4), (6, 3, 1, 192, 320, 1),
]


def effnet_divisible(v, divisor=8, min_value=None):
    """_make_divisible, efficientnet.py:181-194."""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def same_pad(x, k, stride, dilation=1):
    """padding='SAME' of TensorLayerX's torch backend [TLX-recalled]: TensorFlow's rule, odd unit bottom / right."""
    pads = []
    for i in (x.shape[3], x.shape[2]):          # F.pad order: W first
        total = max(0, (-(-i // stride) - 1) * stride + dilation * (k - 1) + 1 - i)
        pads += [total // 2, total - total // 2]
    return F.pad(x, pads)


def _effnet_cna(p, pre, x, stride, groups, silu, eps):
    """ConvNormActivation efficientnet.py:92-125: GroupConv2d('SAME', bias) -> BatchNorm2d -> SiLU (:13-18) or nothing."""
    w = _t(p, pre + ".0.filters")
    y = F.conv2d(same_pad(x, w.shape[-1], stride), w, _t(p, pre + ".0.biases"), stride, 0, 1, groups)
    y = bn(p, pre + ".1", y, eps)
    return y * torch.sigmoid(y) if silu else y


def efficientnet(p, x, arch="efficientnet_b0"):
    """EfficientNet.forward efficientnet.py:426-431; MBConv.forward :302-307; SqueezeExcitation :169-178."""
    wm, dm, eps = EFFNET_ARCH[arch]
    adj = lambda c, m: effnet_divisible(c * m, 8)                              # MBConvConfig.adjust_channels :218-222
    x = _effnet_cna(p, "features.0", x, 2, 1, True, eps)                      # :354-363  3x3 / 2
    out_last = None
    for si, (t, k, s, cin, cout, n) in enumerate(EFFNET_STAGES):
        cin, cout, n = adj(cin, wm), adj(cout, wm), int(math.ceil(n * dm))   # :213-215, :225-226
        for bi in range(n):
            pre = f"features.{si + 1}.{bi}.block"
            in_c, stride = (cin, s) if bi == 0 else (cout, 1)                  # :372-374
            expanded = adj(in_c, t)                                            # :248-249
            y, idx = x, 0
            if expanded != in_c:                                               # :250-259 expand 1x1
                y = _effnet_cna(p, f"{pre}.0", y, 1, 1, True, eps)
                idx = 1
            y = _effnet_cna(p, f"{pre}.{idx}", y, stride, expanded, True, eps)   # :261-271 depthwise k x k
            se = f"{pre}.{idx + 1}"                                            # :273-282 squeeze = max(1, in_c // 4)
            sc = F.adaptive_avg_pool2d(y, 1)
            sc = F.conv2d(sc, _t(p, se + ".fc1.filters"), _t(p, se + ".fc1.biases"))
            sc = sc * torch.sigmoid(sc)
            sc = torch.sigmoid(F.conv2d(sc, _t(p, se + ".fc2.filters"), _t(p, se + ".fc2.biases")))
            y = sc * y                                                         # :176-178
            y = _effnet_cna(p, f"{pre}.{idx + 2}", y, 1, 1, False, eps)        # :284-293 project, no activation
            x = y + x if (stride == 1 and in_c == cout) else y                 # :303-306 (StochasticDepth = identity in eval)
        out_last = cout
    x = _effnet_cna(p, f"features.{len(EFFNET_STAGES) + 1}", x, 1, 1, True, eps)   # :387-397 1x1 -> 4 * out_last
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)                                # :428-429
    return linear(p, "classifier.1", x)                                       # :430 (Dropout = identity)


# ---------------------------------------------------------------------------------------------
# ResNeSt — models/classification/resnest.py
# ---------------------------------------------------------------------------------------------
RESNEST_ARCH = {   # resnest.py:692-735: layers, radix, groups, stem_width, avd_first  (deep_stem, avd, avg_down always on)
    "resnest50_fast_1s1x64d": ([3, 4, 6, 3], 1, 1, 32, True),
    "resnest50": ([3, 4, 6, 3], 2, 1, 32, False),
    "resnest101": ([3, 4, 23, 3], 2, 1, 64, False),
}


def _rs_convbn(p, pre, x, stride=1, groups=1, relu=False):
    """ConvBNLayer.forward resnest.py:48-51 (padding (k-1)//2 :32, no bias :35, BatchNorm(act) :39-45)."""
    k = _t(p, pre + "._conv.filters").shape[-1]
    y = bn(p, pre + ".batch_norm", conv(p, pre + "._conv", x, stride, (k - 1) // 2, 1, groups))
    return F.relu(y) if relu else y


def _rs_splat(p, pre, x, radix, cardinality):
    """SplatConv.forward resnest.py:147-166; rSoftmax.forward :65-81."""
    x = _rs_convbn(p, pre + ".conv1", x, 1, cardinality * radix, True)        # :148  3x3, groups = cardinality * radix
    B, rc = x.shape[0], x.shape[1]
    if radix > 1:
        splited = torch.chunk(x, radix, dim=1)                                # :150-151
        gap = sum(splited[1:], splited[0])                                    # :152
    else:
        gap = x
    gap = F.adaptive_avg_pool2d(gap, 1)                                       # :155
    gap = _rs_convbn(p, pre + ".conv2", gap, 1, cardinality, True)            # :156  1x1, groups = cardinality
    att = conv(p, pre + ".conv3", gap, 1, 0, 1, cardinality)                  # :157  1x1, no bias, no norm
    if radix > 1:                                                             # rSoftmax :69-78
        att = att.reshape(B, cardinality, radix, rc // cardinality // radix).transpose(1, 2)
        att = torch.softmax(att, dim=1).reshape(B, rc, 1, 1)
        attens = torch.chunk(att, radix, dim=1)                               # :160-161
        return sum(a * s_ for a, s_ in zip(attens, splited))                  # :162-163
    return x * torch.sigmoid(att)                                             # :80, :165


def resnest(p, x, arch="resnest50"):
    """ResNeSt.forward resnest.py:679-689; BottleneckBlock.forward :311-328; ResNeStLayer :331-438."""
    layers, radix, card, stem_width, avd_first = RESNEST_ARCH[arch]
    x = _rs_convbn(p, "stem.conv1", x, 2, 1, True)                            # :479-513 deep stem: 3x3/2, 3x3, 3x3
    x = _rs_convbn(p, "stem.conv2", x, 1, 1, True)
    x = _rs_convbn(p, "stem.conv3", x, 1, 1, True)
    x = F.max_pool2d(x, 3, 2, 1)                                              # :527-532
    inplanes = stem_width * 2
    for li, (planes, n) in enumerate(zip((64, 128, 256, 512), layers), start=1):
        for bi in range(n):
            pre = f"layer{li}.layer{li}_bottleneck_{bi}"
            stride = 2 if (bi == 0 and li > 1) else 1                          # :533-676 (no dilation in these archs)
            is_first = li > 1 and bi == 0                                      # layer1 passes is_first=False :545; default True :347
            pool = stride > 1 or is_first                                      # avd pooling condition :313, :316
            short = x
            y = _rs_convbn(p, pre + ".conv1", x, 1, 1, True)                   # :312
            if avd_first and pool:
                y = F.avg_pool2d(y, 3, stride, 1)                              # :313-314 (zero padding counts in the mean)
            y = _rs_splat(p, pre + ".conv2", y, radix, card)                   # :315 (radix >= 1 -> SplatConv :219-233)
            if (not avd_first) and pool:
                y = F.avg_pool2d(y, 3, stride, 1)                              # :316-317
            y = _rs_convbn(p, pre + ".conv3", y)                               # :318
            if stride != 1 or inplanes != planes * 4:                          # :319-323 avg_down shortcut
                short = F.avg_pool2d(short, stride, stride, 0)                 # :271-277 (kernel = stride)
                short = bn(p, pre + ".batch_norm", conv(p, pre + ".conv4", short, 1, 0))
            x = F.relu(short + y)                                              # :324-326
            inplanes = planes * 4
    x = F.adaptive_avg_pool2d(x, 1).reshape(x.shape[0], -1)                   # :686-687
    return linear(p, "out", x)                                                # :688


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


# ---------------------------------------------------------------------------------------------
# Vision Transformer — models/classification/vision_transformer.py
# ---------------------------------------------------------------------------------------------
VIT_CFG = {  # _vision_transformer, vision_transformer.py:336-416
    "vit_small_patch16_224": dict(img=224, patch=16, dim=768, depth=8, heads=8, qk_scale=768 ** -0.5, eps=1e-5),
    "vit_base_patch16_224": dict(img=224, patch=16, dim=768, depth=12, heads=12, qk_scale=None, eps=1e-6),
    "vit_base_patch16_384": dict(img=384, patch=16, dim=768, depth=12, heads=12, qk_scale=None, eps=1e-6),
    "vit_base_patch32_384": dict(img=384, patch=32, dim=768, depth=12, heads=12, qk_scale=None, eps=1e-6),
    "vit_large_patch16_224": dict(img=224, patch=16, dim=1024, depth=24, heads=16, qk_scale=None, eps=1e-6),
}


def vit_attention(p, pre, x, heads, scale):
    """Attention.forward, vision_transformer.py:112-123."""
    N, C = x.shape[1:]
    qkv = linear(p, pre + ".qkv", x).reshape((-1, N, 3, heads, C // heads))        # :114
    qkv = qkv.permute(2, 0, 3, 1, 4)                                                 # :115
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = q.matmul(k.permute(0, 1, 3, 2)) * scale                                   # :117 (scale AFTER q k^T)
    attn = torch.softmax(attn, dim=-1)                                               # :118
    x = attn.matmul(v).permute(0, 2, 1, 3).reshape((-1, N, C))                       # :120
    return linear(p, pre + ".proj", x)                                               # :121


def vit_block(p, pre, x, heads, scale, eps):
    """Block.forward, vision_transformer.py:172-175 (drop_path = Identity at rate 0, :157)."""
    x = x + vit_attention(p, pre + ".attn", layernorm(p, pre + ".norm1", x, eps), heads, scale)
    h = linear(p, pre + ".mlp.fc1", layernorm(p, pre + ".norm2", x, eps))            # Mlp.forward :81-87
    h = F.gelu(h, approximate="none")
    return x + linear(p, pre + ".mlp.fc2", h)


def vit(p, x, arch="vit_base_patch16_224"):
    """VisionTransformer.forward, vision_transformer.py:318-333."""
    c = VIT_CFG[arch]
    B = x.shape[0]
    assert x.shape[2] == c["img"] and x.shape[3] == c["img"]                         # :217-219
    t = conv(p, "patch_embed.proj", x, c["patch"], 0).flatten(2, 3).permute(0, 2, 1)  # PatchEmbed.forward :206-211
    cls = _t(p, "cls_token").expand((B, -1, -1))                                     # :321
    t = torch.cat((cls, t), dim=1) + _t(p, "pos_embed")                              # :322-323
    scale = c["qk_scale"] or (c["dim"] // c["heads"]) ** -0.5                        # :103
    for i in range(c["depth"]):
        t = vit_block(p, f"blocks.{i}", t, c["heads"], scale, c["eps"])
    t = layernorm(p, "norm", t, c["eps"])                                            # :327
    return linear(p, "head", t[:, 0])                                                # :328, :332


# ---------------------------------------------------------------------------------------------
# Swin Transformer — models/classification/swin_transformer.py (Paddle-only file in the reference:
# restated from the text, cannot be imported here)
# ---------------------------------------------------------------------------------------------
SWIN_CFG = {  # _swin_transformer, swin_transformer.py:628-650
    "swintransformer_tiny_patch4_window7_224": dict(img=224, dim=96, depths=[2, 2, 6, 2], heads=[3, 6, 12, 24], ws=7),
    "swintransformer_small_patch4_window7_224": dict(img=224, dim=96, depths=[2, 2, 18, 2], heads=[3, 6, 12, 24], ws=7),
    "swintransformer_base_patch4_window7_224": dict(img=224, dim=128, depths=[2, 2, 18, 2], heads=[4, 8, 16, 32], ws=7),
    "swintransformer_base_patch4_window12_384": dict(img=384, dim=128, depths=[2, 2, 18, 2], heads=[4, 8, 16, 32], ws=12),
    "swintransformer_large_patch4_window7_224": dict(img=224, dim=192, depths=[2, 2, 18, 2], heads=[6, 12, 24, 48], ws=7),
    "swintransformer_large_patch4_window12_384": dict(img=384, dim=192, depths=[2, 2, 18, 2], heads=[6, 12, 24, 48], ws=12),
}
LN_EPS = 1e-5  # nn.LayerNorm default [TLX-recalled]; swin passes no epsilon (swin_transformer.py:258,279)


def swin_window_partition(x, ws):
    """window_partition, swin_transformer.py:85-99."""
    B, H, W, C = x.shape
    x = x.reshape([B, H // ws, ws, W // ws, ws, C])
    return x.permute(0, 1, 3, 2, 4, 5).reshape([-1, ws, ws, C])


def swin_window_reverse(windows, ws, H, W, C):
    """window_reverse, swin_transformer.py:102-116."""
    x = windows.reshape([-1, H // ws, W // ws, ws, ws, C])
    return x.permute(0, 1, 3, 2, 4, 5).reshape([-1, H, W, C])


def swin_relative_position_index(ws):
    """WindowAttention.__init__, swin_transformer.py:146-158."""
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij"))
    cf = torch.flatten(coords, 1)
    rel = (cf.unsqueeze(2) - cf.unsqueeze(1)).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def swin_attn_mask(H, W, ws, shift):
    """SwinTransformerBlock.__init__, swin_transformer.py:288-305 (0 / -100.0, not -inf)."""
    img_mask = torch.zeros((1, H, W, 1))
    cnt = 0
    for h in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for w in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img_mask[:, h, w, :] = cnt
            cnt += 1
    mw = swin_window_partition(img_mask, ws).reshape([-1, ws * ws])
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return -100.0 * (am != 0).float()


def swin_window_attention(p, pre, x, heads, ws, mask):
    """WindowAttention.forward, swin_transformer.py:192-229."""
    B_, N, C = x.shape
    qkv = linear(p, pre + ".qkv", x).reshape([B_, N, 3, heads, C // heads]).permute(2, 0, 3, 1, 4)  # :194-196
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * ((C // heads) ** -0.5)                                                   # :202 scale BEFORE q k^T
    attn = torch.matmul(q, k.permute(0, 1, 3, 2))                                    # :203
    index = swin_relative_position_index(ws).reshape([-1])                           # :205
    bias = torch.index_select(_t(p, pre + ".relative_position_bias_table"), 0, index)
    bias = bias.reshape([ws * ws, ws * ws, -1]).permute(2, 0, 1)                     # :208-212
    attn = attn + bias.unsqueeze(0)                                                  # :213
    if mask is not None:                                                             # :216-220
        nW = mask.shape[0]
        attn = attn.reshape([B_ // nW, nW, heads, N, N]) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.reshape([-1, heads, N, N])
    attn = torch.softmax(attn, dim=-1)
    x = torch.matmul(attn, v).permute(0, 2, 1, 3).reshape([B_, N, C])               # :225-226
    return linear(p, pre + ".proj", x)


def swin_block(p, pre, x, H, W, heads, ws, shift):
    """SwinTransformerBlock.forward, swin_transformer.py:310-337 (window/shift clamp :274-276)."""
    if min(H, W) <= ws:
        shift, ws = 0, min(H, W)
    B, L, C = x.shape
    assert L == H * W, "input feature has wrong size"
    shortcut = x
    x = layernorm(p, pre + ".norm1", x, LN_EPS).reshape([B, H, W, C])
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = swin_window_partition(x, ws).reshape([-1, ws * ws, C])
    mask = swin_attn_mask(H, W, ws, shift) if shift > 0 else None
    aw = swin_window_attention(p, pre + ".attn", xw, heads, ws, mask).reshape([-1, ws, ws, C])
    x = swin_window_reverse(aw, ws, H, W, C)
    if shift > 0:
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    x = shortcut + x.reshape([B, H * W, C])                                          # :334
    h = F.gelu(linear(p, pre + ".mlp.fc1", layernorm(p, pre + ".norm2", x, LN_EPS)), approximate="none")
    return x + linear(p, pre + ".mlp.fc2", h)                                        # :335


def swin_patch_merging(p, pre, x, H, W):
    """PatchMerging.forward, swin_transformer.py:373-391 (the reduction Linear HAS a bias, :369-370)."""
    B, L, C = x.shape
    assert L == H * W, "input feature has wrong size"
    assert H % 2 == 0 and W % 2 == 0, "x size ({}*{}) are not even.".format(H, W)
    x = x.reshape([B, H, W, C])
    x = torch.cat([x[:, 0::2, 0::2, :], x[:, 1::2, 0::2, :], x[:, 0::2, 1::2, :], x[:, 1::2, 1::2, :]], -1)
    x = layernorm(p, pre + ".norm", x.reshape([B, H * W // 4, 4 * C]), LN_EPS)
    return linear(p, pre + ".reduction", x)


def swin(p, x, arch="swintransformer_base_patch4_window7_224"):
    """SwinTransformer.forward, swin_transformer.py:601-616."""
    c = SWIN_CFG[arch]
    x = conv(p, "patch_embed.proj", x, 4, 0)                                         # PatchEmbed.forward :498-504
    x = layernorm(p, "patch_embed.norm", x.flatten(2).permute(0, 2, 1), LN_EPS)
    if "absolute_pos_embed" in p:                                                    # ape=True, :561-565, :603-604
        x = x + _t(p, "absolute_pos_embed")
    res = c["img"] // 4
    for li, (depth, heads) in enumerate(zip(c["depths"], c["heads"])):
        H = W = res // 2 ** li
        for bi in range(depth):                                                      # BasicLayer :427-435, 446-451
            x = swin_block(p, f"layers.{li}.blocks.{bi}", x, H, W, heads, c["ws"], 0 if bi % 2 == 0 else c["ws"] // 2)
        if li < len(c["depths"]) - 1:
            x = swin_patch_merging(p, f"layers.{li}.downsample", x, H, W)
    x = layernorm(p, "norm", x, LN_EPS)                                              # :608
    x = F.adaptive_avg_pool1d(x.permute(0, 2, 1), 1).flatten(1)                      # :609-610
    return linear(p, "head", x)                                                      # :615


# ---------------------------------------------------------------------------------------------
# MobileNetV1 — models/classification/mobilenetv1.py
# ---------------------------------------------------------------------------------------------
MOBILENETV1_PLAN = [(32, 64, 1), (64, 128, 2), (128, 128, 1), (128, 256, 2), (256, 256, 1), (256, 512, 2)] + \
                   [(512, 512, 1)] * 5 + [(512, 1024, 2), (1024, 1024, 1)]        # (c1, c2, stride), :136-244


def _cna(p, pre, x, stride, padding, groups=1):
    """ConvNormActivation = GroupConv2d (no bias) + BatchNorm2d + ReLU, mobilenetv1.py:45-65."""
    return F.relu(bn(p, pre + ".1", conv(p, pre + ".0", x, stride, padding, 1, groups)))


def mobilenetv1(p, x, scale=1.0, num_classes=1000, with_pool=True):
    """MobileNetV1.forward, mobilenetv1.py:254-262; DepthwiseSeparable.forward :99-102."""
    x = _cna(p, "conv1", x, 2, 1)                                                    # :128-135, :255
    for i, (c1, c2, s) in enumerate(MOBILENETV1_PLAN):
        x = _cna(p, f"dwsl.{i}._depthwise_conv", x, s, 1, groups=int(c1 * scale))     # :79-88
        x = _cna(p, f"dwsl.{i}._pointwise_conv", x, 1, 0)                             # :89-97
    if with_pool:
        x = F.adaptive_avg_pool2d(x, 1)                                              # :258
    if num_classes > 0:
        x = linear(p, "fc", x.reshape(x.shape[0], -1))                               # :260-261
    return x


# ---------------------------------------------------------------------------------------------
# DarkNet-53 / YOLOv3 neck + head — models/detection/backbones/darknet.py, models/detection/yolov3.py
# ---------------------------------------------------------------------------------------------
def _cbl(p, pre, x, stride=1, padding=0):
    """ConvBNLayer.forward: conv (no bias) -> BatchNorm -> LeakyReLU(0.1), darknet.py:54-58."""
    return F.leaky_relu(bn(p, pre + ".batch_norm", conv(p, pre + ".conv", x, stride, padding)), 0.1)


def _dark_basic(p, pre, x):
    """BasicBlock.forward, darknet.py:155-159: inputs + conv2(conv1(inputs))."""
    return x + _cbl(p, pre + ".conv2", _cbl(p, pre + ".conv1", x), 1, 1)


def darknet53(p, x, pre="", return_idx=(2, 3, 4)):
    """DarkNet.forward, darknet.py:299-312; stages [1,2,8,8,4] :217; list children named <attr>_<i>."""
    out = _cbl(p, pre + "conv0", x, 1, 1)
    out = _cbl(p, pre + "downsample0.conv_bn_layer", out, 2, 1)
    blocks = []
    for i, n in enumerate([1, 2, 8, 8, 4]):
        b = f"{pre}darknet_conv_block_list_{i}"
        out = _dark_basic(p, b + ".basicblock0", out)                                 # Blocks.forward :208-211
        for j in range(n - 1):
            out = _dark_basic(p, f"{b}.res_blocks.{j}", out)
        if i in return_idx:
            blocks.append(out)
        if i < 4:
            out = _cbl(p, f"{pre}downsample_list_{i}.conv_bn_layer", out, 2, 1)
    return blocks


def mobilenet_det(p, x, pre="", scale=1, feature_maps=(4, 6, 13), extra_block_filters=None):
    """MobileNet.forward of the detection backbone, detection/backbones/mobilenet_v1.py:233-240; ConvBNLayer.forward
    :41-49 (conv without bias, BatchNorm, relu — relu6 in the extra blocks :121-130); DepthwiseSeparable.forward :94-97;
    cfgs :205-215; list children are named dwsl_<i> / extra_blocks_<i>."""
    def cbl(name, x, stride, padding, groups=1, act=F.relu):
        return act(bn(p, name + ".my_batch_norm", conv(p, name + "._conv", x, stride, padding, 1, groups)))
    cfgs = [[32, 64, 1], [64, 128, 2], [128, 128, 1], [128, 256, 2], [256, 256, 1], [256, 512, 2],
            *[[512, 512, 1]] * 5, [512, 1024, 2], [1024, 1024, 1]]
    outs = []
    y = cbl(pre + "conv1", x, 2, 1)
    for i, (ci, co, s_) in enumerate(cfgs):
        y = cbl(f"{pre}dwsl_{i}._depthwise_conv", y, s_, 1, groups=int(ci * scale))
        y = cbl(f"{pre}dwsl_{i}._pointwise_conv", y, 1, 0)
        if i + 1 in feature_maps:
            outs.append(y)
    for i, _ in enumerate(extra_block_filters or []):
        y = cbl(f"{pre}extra_blocks_{i}.pointwise_conv", y, 1, 0, act=F.relu6)
        y = cbl(f"{pre}extra_blocks_{i}.normal_conv", y, 2, 1, act=F.relu6)
        if len(cfgs) + i + 1 in feature_maps:
            outs.append(y)
    return outs


def detr_mha(p, pre, query, key, value, num_heads, attn_mask=None, need_weights=True):
    """MultiHeadAttention.forward, detection/detr.py:1003-1062: sequence-first (L, B, D) inputs, packed in_proj applied
    slice by slice (:1010-1020), q scaled before q k^T (:1022), + attn_mask (:1038-1039), softmax, @ v, out_proj,
    weights averaged over the heads (:1054-1060)."""
    D = query.shape[-1]
    hd = D // num_heads
    W, b = p[pre + "in_proj_weight"], p[pre + "in_proj_bias"]
    T, B, S = query.shape[0], query.shape[1], key.shape[0]
    WQ = torch.matmul(query, W[:D].t()) + b[:D]
    WK = torch.matmul(key, W[D:2 * D].t()) + b[D:2 * D]
    WV = torch.matmul(value, W[2 * D:].t()) + b[2 * D:]
    WQ = (WQ * float(hd) ** -0.5).reshape(T, B * num_heads, hd).permute(1, 0, 2)
    WK = WK.reshape(S, B * num_heads, hd).permute(1, 0, 2)
    WV = WV.reshape(S, B * num_heads, hd).permute(1, 0, 2)
    w = torch.matmul(WQ, WK.transpose(-1, -2))
    if attn_mask is not None:
        w = w + attn_mask
    w = torch.softmax(w, dim=-1)
    o = torch.matmul(w, WV).permute(1, 0, 2).reshape(T, B, D)
    o = torch.matmul(o, p[pre + "out_proj_weight"].t()) + p[pre + "out_proj_bias"]
    if need_weights:
        return o, w.reshape(B, num_heads, T, S).mean(dim=1)
    return o


def yolov3_neck(p, feats, pre="neck."):
    """YOLOv3FPN.forward, yolov3.py:239-258; YoloDetBlock.forward :180-183."""
    X = feats[::-1]
    outs, route = [], None
    for i, x in enumerate(X):
        if i > 0:
            x = torch.cat([route, x], dim=1)                                         # :246
        b = f"{pre}yolo_blocks_{i}"
        for j, (fs) in enumerate([1, 3, 1, 3, 1]):                                    # conv_def :143-149
            x = _cbl(p, f"{b}.conv_module.{j}", x, 1, (fs - 1) // 2)
        route = x
        outs.append(_cbl(p, f"{b}.tip", route, 1, 1))                                # :182
        if i < len(X) - 1:
            route = _cbl(p, f"{pre}routes_{i}", route, 1, 0)                          # :252-253
            route = F.interpolate(route, scale_factor=2.0)                           # :254 (nearest)
    return outs


def yolov3(p, x):
    """YOLOv3.forward up to the raw head maps, yolov3.py:51-68, 352."""
    body = darknet53(p, x, "backbone.")
    neck = yolov3_neck(p, body, "neck.")
    head = [conv(p, f"yolo_head.yolo_outputs_{i}", f) for i, f in enumerate(neck)]
    return body, neck, head


# ---------------------------------------------------------------------------------------------
# MobileNetV2 / V3 — models/classification/mobilenetv2.py, mobilenetv3.py (Paddle-only files: restated
# from the text), ops/ops_fusion.py:11-48, utils/common_func.py:1-16
# ---------------------------------------------------------------------------------------------
def make_divisible(v, divisor=8, min_value=None):
    """_make_divisible, utils/common_func.py:1-16."""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def _cna_act(p, pre, x, k, stride, groups, act, eps=BN_EPS):
    """ConvNormActivation (ops/ops_fusion.py:31-48): conv(pad=(k-1)//2, no bias) + BN + optional activation."""
    y = bn(p, pre + ".1", conv(p, pre + ".0", x, stride, (k - 1) // 2, 1, groups), eps)
    return act(y) if act is not None else y


def mobilenetv2(p, x, scale=1.0):
    """MobileNetV2.forward, mobilenetv2.py:102-109; features built :76-93; InvertedResidual.forward :36-40."""
    setting = [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1], [6, 160, 3, 2], [6, 320, 1, 1]]
    inp = make_divisible(32 * scale, 8)
    x = _cna_act(p, "features.0", x, 3, 2, 1, F.relu6)
    fi = 1
    for t, c, n, s in setting:
        oup = make_divisible(c * scale, 8)
        for i in range(n):
            stride = s if i == 0 else 1
            hidden = int(round(inp * t))
            pre = f"features.{fi}.conv"
            h, li = x, 0
            if t != 1:
                h = _cna_act(p, f"{pre}.{li}", h, 1, 1, 1, F.relu6)
                li += 1
            h = _cna_act(p, f"{pre}.{li}", h, 3, stride, hidden, F.relu6)
            h = bn(p, f"{pre}.{li + 2}", conv(p, f"{pre}.{li + 1}", h))
            x = x + h if (stride == 1 and inp == oup) else h
            inp = oup
            fi += 1
    x = _cna_act(p, f"features.{fi}", x, 1, 1, 1, F.relu6)
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)
    return linear(p, "classifier.1", x)


MBV3_SMALL = [(16, 3, 16, 16, True, 'relu', 2), (16, 3, 72, 24, False, 'relu', 2), (24, 3, 88, 24, False, 'relu', 1),
              (24, 5, 96, 40, True, 'hardswish', 2), (40, 5, 240, 40, True, 'hardswish', 1),
              (40, 5, 240, 40, True, 'hardswish', 1), (40, 5, 120, 48, True, 'hardswish', 1),
              (48, 5, 144, 48, True, 'hardswish', 1), (48, 5, 288, 96, True, 'hardswish', 2),
              (96, 5, 576, 96, True, 'hardswish', 1), (96, 5, 576, 96, True, 'hardswish', 1)]   # mobilenetv3.py:209-221
MBV3_LARGE = [(16, 3, 16, 16, False, 'relu', 1), (16, 3, 64, 24, False, 'relu', 2), (24, 3, 72, 24, False, 'relu', 1),
              (24, 5, 72, 40, True, 'relu', 2), (40, 5, 120, 40, True, 'relu', 1), (40, 5, 120, 40, True, 'relu', 1),
              (40, 3, 240, 80, False, 'hardswish', 2), (80, 3, 200, 80, False, 'hardswish', 1),
              (80, 3, 184, 80, False, 'hardswish', 1), (80, 3, 184, 80, False, 'hardswish', 1),
              (80, 3, 480, 112, True, 'hardswish', 1), (112, 3, 672, 112, True, 'hardswish', 1),
              (112, 5, 672, 160, True, 'hardswish', 2), (160, 5, 960, 160, True, 'hardswish', 1),
              (160, 5, 960, 160, True, 'hardswish', 1)]                                        # :253-269


def mobilenetv3(p, x, config, scale=1.0):
    """MobileNetV3.forward, mobilenetv3.py:170-180; InvertedResidual.forward :111-121; SqueezeExcitation :47-56;
    every BatchNorm has epsilon 1e-3 (:148)."""
    EPS = 1e-3
    adj = lambda c: make_divisible(c * scale, 8)
    x = _cna_act(p, "conv", x, 3, 2, 1, F.hardswish, EPS)
    for i, (cin, k, cexp, cout, use_se, actn, stride) in enumerate(config):
        cin, cexp, cout = adj(cin), adj(cexp), adj(cout)
        act = F.relu if actn == 'relu' else F.hardswish
        pre = f"blocks.{i}"
        identity = x
        if cin != cexp:
            x = _cna_act(p, pre + ".expand_conv", x, 1, 1, 1, act, EPS)
        x = _cna_act(p, pre + ".bottleneck_conv", x, k, stride, cexp, act, EPS)
        if use_se:
            s = F.adaptive_avg_pool2d(x, 1)
            s = F.relu(conv(p, pre + ".mid_se.fc1", s))
            s = F.hardsigmoid(conv(p, pre + ".mid_se.fc2", s))
            x = s * x
        x = _cna_act(p, pre + ".linear_conv", x, 1, 1, 1, None, EPS)
        if stride == 1 and cin == cout:
            x = identity + x
    x = _cna_act(p, "lastconv", x, 1, 1, 1, F.hardswish, EPS)
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)
    x = F.hardswish(linear(p, "classifier.0", x))
    return linear(p, "classifier.3", x)
