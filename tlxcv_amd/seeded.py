"""Seeded, numpy-only weight and input recipes.

There is no network for checkpoints, so benches and parity tests fill models from these explicit
recipes: every value comes from `numpy.random.default_rng(seed)` (never torch's RNG), keyed by the
parameter's dotted attribute path, so the same dictionary loads into the engine's model, into the
CPU oracle, and — in the development container — into the reference's own model file.

The recipes are chosen so that activations keep O(1) magnitude through deep residual stacks (so the
fp16 throughput mode never saturates) while every BatchNorm statistic, bias and affine parameter
is non-trivial (so a parity test would catch a dropped or mis-ordered term).
"""
import numpy as np


def image_batch(n, seed=0, hw=224, c=3):
    """Synthetic ImageNet-shaped batch, NCHW fp32, roughly the range left by the demo's Normalize
    (demo/image_classification/predict.py:25): unit gaussian noise plus, per image and channel, a
    smooth plane wave of random frequency / phase / amplitude and a random DC offset, so different
    images drive a random-weight network to visibly different logits."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, c, hw, hw), dtype=np.float32)
    yy, xx = np.meshgrid(np.arange(hw, dtype=np.float32), np.arange(hw, dtype=np.float32), indexing="ij")
    fy = rng.uniform(-0.2, 0.2, (n, c, 1, 1)).astype(np.float32)
    fx = rng.uniform(-0.2, 0.2, (n, c, 1, 1)).astype(np.float32)
    ph = rng.uniform(0, 2 * np.pi, (n, c, 1, 1)).astype(np.float32)
    amp = rng.uniform(0.5, 2.0, (n, c, 1, 1)).astype(np.float32)
    dc = rng.uniform(-1.0, 1.0, (n, c, 1, 1)).astype(np.float32)
    x += amp * np.sin(fy * yy + fx * xx + ph) + dc
    return np.ascontiguousarray(x, dtype=np.float32)


def _is_last_bn_of_block(name):
    # ResNet: bn3 of a bottleneck / bn2 of a basic block / the downsample BN feed the residual add;
    # DarkNet: the BatchNorm of BasicBlock.conv2 (darknet.py:142-153) does.
    # MobileNetV2/V3: the BatchNorm of the linear projection (mobilenetv2.py:30-33, mobilenetv3.py:107-110).
    # ResNeSt: conv3's BatchNorm and the shortcut BatchNorm of a bottleneck (resnest.py:258-309).
    if "_bottleneck_" in name and name.rsplit("_bottleneck_", 1)[1].split(".", 1)[1:] in (["conv3.batch_norm"], ["batch_norm"]):
        return True
    return name.endswith(("bn3", "downsample.1", "conv2.batch_norm", "linear_conv.1")) or name.endswith(".bn2")


def fill(shapes, seed):
    """shapes: ordered {dotted name: shape}.  Returns {name: float32 array} by these rules:
      *.filters (conv OIHW)      N(0, sqrt(2/fan_in))
      *.weights (Linear in,out)  N(0, 0.02) clipped at 2 sigma         (vision_transformer.py:17)
      *.biases                   N(0, 0.02)
      *.gamma                    U(0.8, 1.2); U(0.2, 0.4) for the BN that feeds a residual add
      *.beta                     N(0, 0.05)
      *.moving_mean              N(0, 0.05)
      *.moving_var               U(0.8, 1.2)
      pos_embed / cls_token / relative_position_bias_table   N(0, 0.02) clipped at 2 sigma
      *_weight (attention projections, out x in)   N(0, sqrt(1/fan_in));   *_bias   N(0, 0.02)
    """
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        leaf = name.rsplit(".", 1)[-1]
        owner = name.rsplit(".", 1)[0] if "." in name else ""
        if leaf == "filters":
            fan_in = int(np.prod(shape[1:]))
            a = rng.standard_normal(shape, dtype=np.float32) * np.float32(np.sqrt(2.0 / fan_in))
        elif leaf == "weights":
            a = np.clip(rng.standard_normal(shape, dtype=np.float32), -2, 2) * np.float32(0.02)
        elif leaf == "biases":
            a = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.02)
        elif leaf == "gamma":
            lo, hi = (0.2, 0.4) if _is_last_bn_of_block(owner) else (0.8, 1.2)
            a = rng.uniform(lo, hi, shape).astype(np.float32)
        elif leaf in ("beta", "moving_mean"):
            a = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.05)
        elif leaf == "moving_var":
            a = rng.uniform(0.8, 1.2, shape).astype(np.float32)
        elif leaf in ("pos_embed", "cls_token", "relative_position_bias_table", "absolute_pos_embed"):
            a = np.clip(rng.standard_normal(shape, dtype=np.float32), -2, 2) * np.float32(0.02)
        elif leaf.endswith("_weight") and len(shape) == 2:      # attention projections stored (out, in): detr.py:975-995
            a = rng.standard_normal(shape, dtype=np.float32) * np.float32(np.sqrt(1.0 / shape[1]))
        elif leaf.endswith("_bias"):
            a = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.02)
        else:
            raise KeyError(f"seeded.fill: no rule for parameter {name!r}")
        out[name] = np.ascontiguousarray(a, dtype=np.float32)
    return out


def shapes_of(module):
    """Ordered {name: shape} of a torch-style module's floating parameters and buffers."""
    derived = ("attn_mask", "relative_position_bias", "relative_position_index")   # computed, not learned
    return {k: tuple(v.shape) for k, v in module.state_dict().items()
            if v.is_floating_point() and k.rsplit(".", 1)[-1] not in derived}
