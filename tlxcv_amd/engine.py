"""Functional host layer over the C-ABI: torch tensors in, torch tensors out, HIP kernels inside.

PyTorch-ROCm is used here for device memory, streams and nothing else: every function below
hands `tensor.data_ptr()` and the current HIP stream to libtlxmi.so.  Activations are NHWC
(`(N, H, W, C)` contiguous, or `(rows, C)` token matrices); fp16 is the throughput dtype, fp32 the
parity dtype.  There is deliberately no CPU branch: a CPU tensor raises.
"""
import ctypes as C
import threading

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_HARDSIGMOID, ACT_HARDSWISH, ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_RELU6,
                   ACT_SIGMOID, ACT_SILU, EPI_RES_AFTER_ACT, F16, F32)

EPI_RES_BCAST_N = 2

_precision = torch.float16

# Host-side A/B switches (tools/ only; no environment variable is read on the product path).
_options = {"splitk": True,       # classifier heads: K slices side by side (tlxmi_linear_splitk)
            "attn_comb": True,    # Swin attention with the pre-summed bias + mask table (tlxmi_attention_comb)
            "seams": True,        # block-to-block seams of the bottleneck families as one launch (tlxmi_bottleneck_seam); off = the
                                  # expand conv and the next block's reduce conv as two launches (the A/B and the parity tests' other arm)
            "seam256": True,      # bottleneck seams with a 256-channel conv3 input (ResNet-50 layer3, 14 x 14) fused too
            "two_streams": True,  # large batches as two half batches on two HIP streams (two_streams(), below)
            "conv_splitk": True,  # convs with few pixels and a long K on K slices (tlxmi_conv2d_splitk)
            "patch_linear": True, # ViT patch embedding as one Linear over all token rows (tlxmi_patchify + the persistent GEMM); off = the
                                  # space-to-depth implicit GEMM writing rows 1.. of each image (the A/B and the parity tests' other arm)
            "patch_embed4": True, # Swin patch embedding (conv 4 x 4 / 4 + LayerNorm) as one pass over the NCHW image (tlxmi_patch_embed4, fp16)
            "mlp_seam": True,     # Swin stage 1 (128 -> 512 -> 128): fc1 + GELU + fc2 + residual as one launch (tlxmi_mlp_seam), the hidden map stays on chip
            "lnfold": True,       # LayerNorm folded AROUND the Linear layers of a transformer block (fp16, round 5): proj / fc2 / the patch embedding
                                  # emit the row statistics of the residual stream from their epilogues (tlxmi_linear_stats), qkv / fc1 apply
                                  # the normalisation in theirs (tlxmi_linear_ln); off = LayerNorm launches + plain Linear layers (the parity
                                  # tests' other arm)
            "lnfold_min_rows": 2048,  # (a number) folded Linear layers need at least this many rows (below, the tiled kernels of the dispatcher win)
            "lnfold_min_rows_one_stream": 12000,  # (a number) ... and this many outside a two-stream forward
            "lnfold_min_c": 256,  # (a number) Swin: fold the LayerNorms of the stages with at least this many channels (tools/ A/B per stage)
            "tail_splitk": False} # Linear layers: the rows of a short last round of 256 x 256 tiles on K slices (_linear_tail): built,
                                  # parity-green, measured a LOSS on the ViT-B/16 forward (10.63 -> 11.61 ms for every K >= 768,
                                  # 10.91 for fc2 only: two more launches + the fp32 partial planes cost more than the idle round)


def set_option(name, value):
    if name not in _options:
        raise KeyError(f"unknown option {name!r}; have {sorted(_options)}")
    _options[name] = int(value) if isinstance(_options[name], int) and not isinstance(_options[name], bool) else bool(value)


def option_value(name):
    """A numeric option (set_option keeps the type the option was declared with)."""
    return _options[name]


def option(name):
    return _options[name]


# Optional per-launch probe (bench.py): when a list is installed, every implicit-GEMM launch appends
# (start_event, end_event, algorithmic_bytes, flops) recorded on the launch stream.
_probe = None


def set_probe(lst):
    global _probe
    _probe = lst


def set_precision(p):
    """'fp16' (throughput: fp16 storage, fp32 accumulate) or 'fp32' (parity: exact fp32)."""
    global _precision
    if p in ("fp16", "float16", torch.float16):
        _precision = torch.float16
    elif p in ("fp32", "float32", torch.float32):
        _precision = torch.float32
    else:
        raise ValueError(f"unknown precision {p!r}")


def precision():
    return _precision



def dt_code(dtype):
    if dtype == torch.float16:
        return F16
    if dtype == torch.float32:
        return F32
    raise RuntimeError(f"tlxcv_amd: unsupported dtype {dtype} (fp16 or fp32 only)")


def vec(dtype):
    return 8 if dtype == torch.float16 else 4


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def need_gpu(t, what="tensor"):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"tlxcv_amd: {what} must be a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(
            f"tlxcv_amd: {what} lives on {t.device}; this engine only runs on an MI355X (HIP) device "
            "and has no CPU path — move the model and inputs to 'cuda'.")
    return t


def to_model_device(inputs, model):
    """Task-boundary upload: host tensors (or the {"images": ...} dict of the detectors, darknet.py:300) handed to a
    model that lives on the GPU are copied there — the reference scripts build their image on the host
    (predict.py:22-29).  A data transfer, not a compute path: a model that is itself on the CPU still raises."""
    p = next(model.parameters(), None)
    if p is None or not p.is_cuda:
        return inputs
    if isinstance(inputs, torch.Tensor):
        return inputs if inputs.is_cuda else inputs.to(p.device, non_blocking=True)
    if isinstance(inputs, dict):
        return {k: to_model_device(v, model) for k, v in inputs.items()}
    if isinstance(inputs, (list, tuple)):
        return type(inputs)(to_model_device(v, model) for v in inputs)
    return inputs


# Two half batches on two HIP streams.  A forward is a chain of launches that are each either MFMA-bound (3x3 convs, the
# Linear layers) or HBM-bound (1x1 convs with their skip, LayerNorm, window plumbing), one workgroup per CU for the big ones:
# every launch ends in a partly filled round while its successor waits.  With the batch cut in two and the halves on two
# streams the hardware fills those tails (and overlaps launches of different kind) with the other half's workgroups: measured,
# hipGraph replay, ResNet-50 batch 256 3.77 -> 3.52 ms (batch 512 7.11 -> 6.43, batch 128 2.17 -> 2.12, batch 64 -1 %), Swin-B
# batch 128 9.14 -> 8.60 ms, VGG-16 batch 64 2.75 -> 2.63 ms, ViT-B/16 batch 256 unchanged (not applied there).  The halves are
# independent (eval-mode forward, no batch statistics): the result is the concatenation, row for row what the halves give alone.
_cache_builds = 0
# Per host THREAD: whether a two-stream forward is being enqueued and which planning hint its launches carry
# (TLXMI_PLAN_SHARED_* in every conv / linear descriptor).  Nothing process-wide is mutated: two threads — or two models —
# may enqueue forwards at the same time without changing each other's tile choices.
_tls = threading.local()


def in_halves():
    """True while a two-stream forward is being enqueued (launch-shape choices that count on the other stream's launches)."""
    return getattr(_tls, "depth", 0) > 0


def plan_flags():
    """Planning bits for the descriptors of the launches enqueued by this thread right now."""
    return getattr(_tls, "plan", 0)


class shared_plan:
    """with shared_plan("half" | "full" | None): the conv / linear launches enqueued by this thread carry TLXMI_PLAN_SHARED_*."""

    def __init__(self, plan):
        self.bits = {"half": _lib.PLAN_SHARED_HALF, "full": _lib.PLAN_SHARED_FULL, None: 0}[plan]

    def __enter__(self):
        self.prev = getattr(_tls, "plan", 0)
        _tls.plan = self.bits
        return self

    def __exit__(self, *exc):
        _tls.plan = self.prev
        return False


def note_cache_build():
    """A layer built a derived tensor (packed filter, folded BatchNorm, bias / mask table) just now, on the current stream.
    Not while a hipGraph is being captured: the build (and the redo of the two-stream forward it triggers) would be recorded
    into the graph for good — run one eager forward first (bench.py, graph.GraphedForward and the tests do)."""
    global _cache_builds
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        raise RuntimeError("tlxcv_amd: a layer built its packed filter / folded BatchNorm / table during hipGraph capture; "
                           "run one eager forward of the model (same precision, same weights) before capturing")
    _cache_builds += 1


def run_halves(fn, x, plan=None):
    """fn(first half) on the current stream, fn(second half) on the device's side stream, joined; returns the concatenation.
    Derived tensors are built lazily on whichever stream asks first, and the other stream would read them unordered: when a
    build happened during the call (first forward, new weights, new precision) the streams are joined and the two halves are
    done again — the same launches every later call makes, so the first result equals the later ones bit for bit.
    plan: while the halves are enqueued every conv / linear descriptor carries a planning hint (TLXMI_PLAN_SHARED_*: the
    launch shares the device) — "half": tiles priced for half the CUs and no tail splits (ResNet-50 batch 256 -3 %; Swin-B
    +3 %), "full": the device's CU count, no tail splits only (Swin-B batch 128 -1.3 %).  The hint is per call, the side stream
    and the bookkeeping per host thread.  What IS process-wide: the count of derived-tensor builds — two threads that share ONE
    model must not race its first forward (run it once on one thread first); separate models per thread are independent."""
    cur = torch.cuda.current_stream(x.device)
    idx = x.device.index if x.device.index is not None else torch.cuda.current_device()
    sides = getattr(_tls, "side_streams", None)      # per host thread AND device: two threads forwarding on one GPU do not
    if sides is None:                                # serialise through one shared side stream
        sides = _tls.side_streams = {}
    side = sides.get(idx)
    if side is None:
        side = sides[idx] = torch.cuda.Stream(device=x.device)
    n = x.shape[0] // 2
    _tls.depth = getattr(_tls, "depth", 0) + 1
    hint = shared_plan(plan)
    hint.__enter__()
    try:
        for _ in range(3):
            builds = _cache_builds
            side.wait_stream(cur)                 # x is ready for the side stream (and: a redo starts after everything before it)
            y0 = fn(x[:n])                        # (first: lazy builds land on the caller's stream)
            with torch.cuda.stream(side):
                y1 = fn(x[n:])
            cur.wait_stream(side)
            if _cache_builds == builds:
                y1.record_stream(cur)
                return torch.cat((y0, y1), 0)
    finally:
        _tls.depth -= 1
        hint.__exit__()
    return fn(x)


def two_streams(min_batch, plan=None, eager=True):
    """Decorator of a model's forward(self, x): batches of at least `min_batch` (even) images run as run_halves().
    plan: None / "half" / "full" (run_halves), or a callable batch -> one of these.
    eager=False: only while a hipGraph is being captured.  A forward of many tiny launches (MobileNetV3, EfficientNet: 150 - 250
    kernels of a few microseconds) is bound by the host when launched kernel by kernel, and two halves are twice the host work
    (MobileNetV3-small batch 256: 1.4 -> 2.3 ms eager, 1.37 -> 1.25 ms as a graph)."""
    def deco(fwd):
        import functools

        @functools.wraps(fwd)
        def wrapper(self, x, *args, **kwargs):
            if (_options["two_streams"] and not args and not kwargs and isinstance(x, torch.Tensor) and x.is_cuda
                    and x.dim() == 4 and x.shape[0] >= min_batch and x.shape[0] % 2 == 0 and _probe is None
                    and (eager or torch.cuda.is_current_stream_capturing())):
                return run_halves(lambda h: fwd(self, h), x, plan(x.shape[0]) if callable(plan) else plan)
            return fwd(self, x, *args, **kwargs)
        return wrapper
    return deco


def _f32(t):
    if t is None:
        return None
    need_gpu(t, "parameter")
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.detach().to(torch.float32).contiguous()
    return t.detach()


# ---------------------------------------------------------------------------------------------
# layout
# ---------------------------------------------------------------------------------------------
def nchw_to_nhwc(x, dtype=None, cpad=None):
    """(N,C,H,W) contiguous -> (N,H,W,Cpad) `dtype`; extra channels are zero."""
    need_gpu(x, "input")
    dtype = dtype or _precision
    N, Cc, H, W = x.shape
    v = vec(dtype)
    cpad = cpad or (Cc + v - 1) // v * v
    if x.dtype not in (torch.float16, torch.float32):
        x = x.float()
    x = x.contiguous()
    y = torch.empty((N, H, W, cpad), dtype=dtype, device=x.device)
    _lib.call("tlxmi_nchw_to_nhwc", _p(x), dt_code(x.dtype), _p(y), dt_code(dtype), N, Cc, H, W, cpad, _stream())
    return y


def nchw_to_nhwc_s2d(x, b, dtype=None):
    """(N,C,H,W) -> (N,H/b,W/b,pad(b*b*C)): b x b space-to-depth fold, channel (ph*b+pw)*C + c."""
    need_gpu(x, "input")
    dtype = dtype or _precision
    N, Cc, H, W = x.shape
    if H % b or W % b:
        raise RuntimeError(f"space-to-depth needs H, W multiples of {b}, got {H}x{W}")
    v = vec(dtype)
    cpad = (b * b * Cc + v - 1) // v * v
    if x.dtype not in (torch.float16, torch.float32):
        x = x.float()
    x = x.contiguous()
    y = torch.empty((N, H // b, W // b, cpad), dtype=dtype, device=x.device)
    _lib.call("tlxmi_nchw_to_nhwc_s2d", _p(x), dt_code(x.dtype), _p(y), dt_code(dtype), N, Cc, H, W, b, cpad, _stream())
    return y


def patchify(x, ps, lead=0, dtype=None):
    """(N,C,H,W) -> (N, lead + (H/ps)(W/ps), C*ps*ps) patch rows in the order of the flattened conv filter [Cout][C][ps][ps]; the `lead`
    rows in front of each image's patches are zero (tlxmi_patchify)."""
    need_gpu(x, "input")
    dtype = dtype or _precision
    N, Cc, H, W = x.shape
    if ps % 8 or H % ps or W % ps:
        raise RuntimeError(f"patchify needs a patch size that is a multiple of 8 and divides H, W; got {ps} on {H}x{W}")
    if x.dtype not in (torch.float16, torch.float32):
        x = x.float()
    x = x.contiguous()
    y = torch.empty((N, lead + (H // ps) * (W // ps), Cc * ps * ps), dtype=dtype, device=x.device)
    _lib.call("tlxmi_patchify", _p(x), dt_code(x.dtype), _p(y), dt_code(dtype), N, Cc, H, W, ps, lead, _stream())
    return y


def patch_embed4_filter(w_oihw):
    """Conv filter (D, 3, 4, 4) -> the (D, 64) fp16 image tlxmi_patch_embed4 reads: k = 16 c + 4 ky + kx, zeros above 48."""
    w = _f32(w_oihw)
    D = w.shape[0]
    out = torch.zeros((D, 64), dtype=torch.float16, device=w.device)
    out[:, :48] = w.reshape(D, 48).to(torch.float16)
    return out


def patch_embed4(x, w64, bias, gamma, beta, eps, pos=None):
    """(N,3,H,W) fp32 / fp16 -> (N, H/4 * W/4, D) fp16 = LayerNorm(conv4x4/4(x) + bias) (no LayerNorm when gamma is None) [+ pos, the
    (H/4 * W/4, D) absolute position embedding of SwinTransformer(ape=True)], one pass."""
    need_gpu(x, "input")
    if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] % 4 or x.shape[3] % 4:
        raise RuntimeError(f"patch_embed4: a (N, 3, H, W) image with H, W multiples of 4 is expected, got {tuple(x.shape)} (the kernel reads 3 colour planes)")
    if w64.dim() != 2 or w64.shape[1] != 64 or w64.dtype != torch.float16:
        raise RuntimeError("patch_embed4: the filter must be the (D, 64) fp16 image of patch_embed4_filter()")
    if (gamma is None) != (beta is None):
        raise RuntimeError("patch_embed4: gamma and beta go together (both None: no LayerNorm)")
    if x.dtype not in (torch.float16, torch.float32):
        x = x.float()
    x = x.contiguous()
    N, Cc, H, W = x.shape
    D = w64.shape[0]
    y = torch.empty((N, (H // 4) * (W // 4), D), dtype=torch.float16, device=x.device)
    if pos is not None:
        pos = _f32(pos).reshape(-1, D).contiguous()
        if pos.shape[0] != (H // 4) * (W // 4):
            raise RuntimeError(f"patch_embed4: position table of {pos.shape[0]} rows for {(H // 4) * (W // 4)} tokens")
        _lib.call("tlxmi_patch_embed4_pos", _p(x), dt_code(x.dtype), _p(w64), _p(_f32(bias)), _p(_f32(gamma)), _p(_f32(beta)), _p(pos), _p(y), N, H, W, D,
                  C.c_float(eps), _stream())
        return y
    _lib.call("tlxmi_patch_embed4", _p(x), dt_code(x.dtype), _p(w64), _p(_f32(bias)), _p(_f32(gamma)), _p(_f32(beta)), _p(y), N, H, W, D,
              C.c_float(eps), _stream())
    return y


_resample_tables = {}


def _resample_table(n_in, n_out, interpolation, device):
    key = (n_in, n_out, interpolation, str(device))
    if key not in _resample_tables:
        import numpy as np
        from .tlx.vision.transforms import resample
        if n_in == n_out:       # PIL skips the pass: identity table
            b = np.stack([np.arange(n_out, dtype=np.int32), np.ones(n_out, dtype=np.int32)], 1)
            k = np.full((n_out, 1), 1 << resample.PRECISION_BITS, dtype=np.int32)
        else:
            b, k = resample.coefficients(n_in, n_out, interpolation)
        _resample_tables[key] = (torch.from_numpy(np.ascontiguousarray(b)).to(device), torch.from_numpy(np.ascontiguousarray(k)).to(device), k.shape[1])
    return _resample_tables[key]


def preprocess_u8(images, size, mean=None, std=None, layout="CHW", dtype=torch.float32, interpolation="bilinear", fold=0):
    """uint8 HWC images (N,H,W,C) on the device -> Resize(size) -> Normalize(mean, std) (or /255 without) -> ToTensor,
    one call (tlxmi_preprocess_u8), bit-identical to the host pipeline of tlx.vision.transforms.  layout 'CHW' -> (N,C,h,w),
    'HWC' -> (N,h,w,C); fold=b -> the b x b space-to-depth NHWC image the stem kernels read ((N,h/b,w/b,pad(b*b*C)))."""
    need_gpu(images, "images")
    if images.dtype != torch.uint8 or images.dim() != 4:
        raise RuntimeError("preprocess_u8: a (N, H, W, C) uint8 tensor is expected")
    images = images.contiguous()
    N, H, W, Cc = images.shape
    oh, ow = (size, size) if isinstance(size, int) else (int(size[0]), int(size[1]))
    xb, xk, kw = _resample_table(W, ow, interpolation, images.device)
    yb, yk, kh = _resample_table(H, oh, interpolation, images.device)
    norm = mean is not None
    def per_channel(v):      # a scalar or length-1 mean / std broadcasts over the channels, as on the host (numpy)
        t = torch.as_tensor(v, dtype=torch.float32).reshape(-1)
        if t.numel() == 1 and Cc > 1:
            t = t.expand(Cc)
        return t.to(images.device).contiguous()
    m = per_channel(mean) if norm else None
    sd = per_channel(std if std is not None else 1.0) if norm else None
    if norm and (m.numel() != Cc or sd.numel() != Cc):
        raise RuntimeError(f"preprocess_u8: mean / std must have 1 or {Cc} entries")
    if fold:
        v = vec(dtype)
        cpad = (fold * fold * Cc + v - 1) // v * v
        out = torch.empty((N, oh // fold, ow // fold, cpad), dtype=dtype, device=images.device)
        lay = 2
    elif layout == "CHW":
        out, lay, cpad = torch.empty((N, Cc, oh, ow), dtype=dtype, device=images.device), 0, 0
    elif layout == "HWC":
        out, lay, cpad = torch.empty((N, oh, ow, Cc), dtype=dtype, device=images.device), 1, 0
    else:
        raise ValueError("layout should be CHW or HWC")
    d = _lib.PreprocDesc(N=N, H=H, W=W, C=Cc, out_h=oh, out_w=ow, kh=kh, kw=kw, out_dtype=dt_code(dtype), layout=lay, fold_b=int(fold),
                         cpad=cpad, normalize=1 if norm else 0)
    ws = torch.empty(_lib.load().tlxmi_preprocess_u8_workspace_bytes(C.byref(d)), dtype=torch.uint8, device=images.device)
    _lib.call("tlxmi_preprocess_u8", C.byref(d), _p(images), _p(xb), _p(xk), _p(yb), _p(yk), _p(m), _p(sd), _p(ws), _p(out), _stream())
    return out


def s2d_filter(w_oihw, b, pad):
    """Re-index an OIHW filter for a b x b space-to-depth input (host side, once):
    W2[o][(ph*b+pw)*C + c][r2][s2] = W[o][c][b*r2 + ph - off][b*s2 + pw - off], zero outside;
    returns (W2, new padding).  The conv then runs with stride/b and output extent cropped by the caller."""
    O, Cc, R, S = w_oihw.shape
    ph_ = pad if isinstance(pad, int) else pad[0]
    pad2 = (ph_ + b - 1) // b
    off = pad2 * b - ph_
    R2, S2 = (R + off + b - 1) // b, (S + off + b - 1) // b
    w2 = torch.zeros((O, b * b * Cc, R2, S2), dtype=torch.float32, device=w_oihw.device)
    w = w_oihw.detach().float()
    for ph in range(b):
        for pw in range(b):
            for r2 in range(R2):
                r = b * r2 + ph - off
                if not 0 <= r < R:
                    continue
                for s2 in range(S2):
                    sx = b * s2 + pw - off
                    if 0 <= sx < S:
                        w2[:, (ph * b + pw) * Cc:(ph * b + pw + 1) * Cc, r2, s2] = w[:, :, r, sx]
    return w2, pad2


def nhwc_to_nchw(x, C_true=None, dtype=None):
    """(N,H,W,ld) -> contiguous (N,C,H,W)."""
    need_gpu(x)
    N, H, W, ld = x.shape
    Cc = C_true or ld
    dtype = dtype or x.dtype
    y = torch.empty((N, Cc, H, W), dtype=dtype, device=x.device)
    _lib.call("tlxmi_nhwc_to_nchw", _p(x), dt_code(x.dtype), ld, _p(y), dt_code(dtype), N, Cc, H, W, _stream())
    return y


# ---------------------------------------------------------------------------------------------
# filters / batch-norm folding
# ---------------------------------------------------------------------------------------------
class PackedFilter:
    """A conv / linear weight in the K-contiguous, padded image tlxmi_conv2d reads."""

    def __init__(self, w_oihw, dtype):
        w = _f32(w_oihw)
        if w.dim() == 2:  # Linear weight [out, in]
            w = w.reshape(w.shape[0], w.shape[1], 1, 1)
        self.Cout, self.Cin, self.R, self.S = w.shape
        self.dtype = dtype
        v = vec(dtype)
        self.Cin_pad = (self.Cin + v - 1) // v * v
        nbytes = _lib.load().tlxmi_packed_filter_bytes(self.Cout, self.Cin, self.R, self.S, dt_code(dtype))
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        _lib.call("tlxmi_pack_filter", _p(w), _p(self.buf), self.Cout, self.Cin, self.R, self.S, dt_code(dtype),
                  _stream())


class PackedGroupFilter:
    """Filter of a grouped convolution (1 < groups < C, resnext.py:30-40): `chunks` block-diagonal packed filters,
    one per launch chunk of tlxmi_group_conv2d."""

    def __init__(self, w_oihw, groups, dtype):
        w = _f32(w_oihw)
        self.Cout, cg, self.R, self.S = w.shape
        self.groups = int(groups)
        self.Cin = cg * self.groups
        self.Cin_pad = self.Cin
        self.dtype = dtype
        lib = _lib.load()
        self.chunks = lib.tlxmi_group_conv_chunks(self.Cin, self.Cout, self.groups, dt_code(dtype))
        if self.chunks <= 0:
            raise NotImplementedError(f"grouped conv {self.Cin}->{self.Cout} in {self.groups} groups: channel counts "
                                      "cannot be merged into 16-byte aligned launch chunks")
        nbytes = lib.tlxmi_packed_group_filter_bytes(self.Cout, self.Cin, self.R, self.S, self.groups, dt_code(dtype))
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        _lib.call("tlxmi_pack_group_filter", _p(w), _p(self.buf), self.Cout, self.Cin, self.R, self.S, self.groups,
                  dt_code(dtype), _stream())


def fold_bn(gamma, beta, mean, var, eps, conv_bias=None):
    """Eval-mode BatchNorm (+ optional conv bias) -> per-channel fp32 (scale, shift)."""
    ref = next(t for t in (gamma, beta, mean, var, conv_bias) if t is not None)
    Cn = ref.numel()
    g, b, m, v, cb = (_f32(t) for t in (gamma, beta, mean, var, conv_bias))
    scale = torch.empty(Cn, dtype=torch.float32, device=ref.device)
    shift = torch.empty(Cn, dtype=torch.float32, device=ref.device)
    _lib.call("tlxmi_fold_bn", _p(g), _p(b), _p(m), _p(v), _p(cb), float(eps), Cn, _p(scale), _p(shift), _stream())
    return scale, shift


# ---------------------------------------------------------------------------------------------
# conv / linear
# ---------------------------------------------------------------------------------------------
_cus = {}


def _pair(v):
    return (v, v) if isinstance(v, int) else (int(v[0]), int(v[1]))


def conv2d(x, pk, stride=1, padding=0, dilation=1, scale=None, shift=None, res=None, act=ACT_NONE,
           act_param=0.0, res_after_act=False, out=None, out_ld=None, y_nstride=0, res_nstride=0,
           res_bcast=False, res_ld=None, out_hw=None, overhang=False, maxpool3s2=False):
    """x (N,H,W,C>=Cin_pad...) NHWC -> y (N,Ho,Wo,Cout).  `out` may be a wider/pre-offset buffer.
    maxpool3s2: fold nn.MaxPool2d(3, 2, 1) into the conv's epilogue (TLXMI_EPI_MAXPOOL_3S2P1) -> (N,Ho/2,Wo/2,Cout);
    returns None — nothing launched — when the library has no fused kernel for this geometry."""
    need_gpu(x, "input")
    N, H, W, ld = x.shape
    if x.dtype != pk.dtype:
        raise RuntimeError(f"conv2d: input dtype {x.dtype} != packed filter dtype {pk.dtype}")
    if ld < pk.Cin_pad:
        raise RuntimeError(f"conv2d: input has {ld} channels, filter expects {pk.Cin} (padded {pk.Cin_pad})")
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    Ho = (H + 2 * ph - dh * (pk.R - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dw * (pk.S - 1) - 1) // sw + 1
    if Ho <= 0 or Wo <= 0:
        raise RuntimeError(f"conv2d: empty output {Ho}x{Wo} for input {H}x{W}")
    if out_hw is not None:      # crop (asymmetric padding of a space-to-depth stem), or one-sided end padding ('SAME' at stride 2)
        Ho, Wo = (int(out_hw[0]), int(out_hw[1])) if overhang else (min(Ho, out_hw[0]), min(Wo, out_hw[1]))
    if maxpool3s2 and (out is not None or res is not None):
        return None
    if out is None:
        out = None if maxpool3s2 else torch.empty((N, Ho, Wo, pk.Cout), dtype=x.dtype, device=x.device)
        out_ld = pk.Cout
    elif out_ld is None:
        out_ld = out.shape[-1]
    d = _lib.ConvDesc(dtype=dt_code(x.dtype), N=N, H=H, W=W, C=pk.Cin_pad, Cout=pk.Cout, R=pk.R, S=pk.S,
                      stride_h=sh, stride_w=sw, pad_h=ph, pad_w=pw, dil_h=dh, dil_w=dw, Ho=Ho, Wo=Wo,
                      x_ld=ld, y_ld=out_ld, res_ld=(res_ld if res_ld is not None else (res.shape[-1] if res is not None else 0)),
                      y_nstride=y_nstride, res_nstride=res_nstride, act=act, act_param=float(act_param),
                      flags=(EPI_RES_AFTER_ACT if res_after_act else 0) | (EPI_RES_BCAST_N if res_bcast else 0)
                      | (_lib.EPI_MAXPOOL_3S2P1 if maxpool3s2 else 0) | plan_flags())
    if res is not None and res.dtype != x.dtype:
        raise RuntimeError("conv2d: residual dtype mismatch")
    if maxpool3s2:
        if not _lib.load().tlxmi_conv2d_maxpool_supported(C.byref(d)):
            return None
        out = torch.empty((N, Ho // 2, Wo // 2, pk.Cout), dtype=x.dtype, device=x.device)
    M = N * Ho * Wo
    splits = 0 if maxpool3s2 else _conv_splits(d, M, pk, x, out_ld, y_nstride, res_nstride, res_bcast)

    def launch():
        if splits:
            # few output pixels, long K (the 7 x 7 stage of a ResNet, small batches): K slices side by side + a reduction
            part = torch.empty((splits, M, pk.Cout), dtype=torch.float32, device=x.device)
            _lib.call("tlxmi_conv2d_splitk", C.byref(d), splits, _p(x), _p(pk.buf), _p(part), _p(scale), _p(shift), _p(res), _p(out), _stream())
        else:
            _lib.call("tlxmi_conv2d", C.byref(d), _p(x), _p(pk.buf), _p(scale), _p(shift), _p(res), _p(out), _stream())
    if _probe is None:
        launch()
        return out
    es = x.element_size()
    alg_bytes = (N * H * W * pk.Cin + (M // 4 if maxpool3s2 else M) * pk.Cout * (2 if res is not None else 1)
                 + pk.Cout * pk.Cin * pk.R * pk.S) * es
    flops = 2 * M * pk.Cout * pk.Cin * pk.R * pk.S
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    _probe.append((e0, e1, alg_bytes, flops, (N, H, W, pk.Cin, pk.Cout, pk.R, sh, res is not None)))
    return out


def _conv_splits(d, M, pk, x, out_ld, y_nstride, res_nstride, res_bcast):
    """Number of K slices for a convolution whose few tiles each walk a long K in sequence while most CUs idle (0: one launch).
    The slices cost a second launch (the reduction) and the partial planes, ~20 us: measured (tools/splitk_conv_check.py) the
    3x3 512 -> 512 convs of ResNet's 7 x 7 stage win up to 32 images (26 tiles x 72 K tiles: 46.9 -> 32.5 us in 8 slices; 4 images
    32.8 -> 26.7) and lose from 128 (98 tiles: 53.7 -> 64.2 us); 36 K tiles (14 x 14, 256 channels) lose at any batch.
    set_option("conv_splitk", False) turns the path off (A/B)."""
    if not _options["conv_splitk"] or y_nstride or res_nstride or res_bcast or out_ld != pk.Cout or pk.Cout < 128:
        return 0
    if not (pk.S == 3 or (pk.R == 1 and pk.S == 1 and (d.stride_h > 1 or d.stride_w > 1))):
        return 0
    es = x.element_size()
    ktiles = pk.R * pk.S * pk.Cin_pad * es // 128
    if ktiles < 64:
        return 0
    idx = x.device.index if x.device.index is not None else torch.cuda.current_device()
    if idx not in _cus:
        _cus[idx] = torch.cuda.get_device_properties(idx).multi_processor_count
    tiles = ((M + 127) // 128) * ((pk.Cout + 255) // 256)
    if tiles * 8 > _cus[idx]:
        return 0
    s_ = min(8, _cus[idx] // max(tiles, 1), ktiles // 8)
    if s_ < 2:
        return 0
    return s_ if _lib.load().tlxmi_conv2d_splitk_supported(C.byref(d), s_) else 0


def bottleneck_seam_supported(K1, N1, N2, dtype):
    return bool(_lib.load().tlxmi_bottleneck_seam_supported(dt_code(dtype), int(K1), int(N1), int(N2)))


def bottleneck_seam(t2, pk3, scale3, shift3, skip, pk1, scale1, shift1, proj=None):
    """One launch for the seam between two bottleneck blocks (tlxmi_bottleneck_seam): y = relu(conv3(t2) * scale3 + shift3 +
    skip); t1 = relu(conv1'(y) * scale1 + shift1).  t2 (N,H,W,K1), skip (N,H,W,N1) -> (y (N,H,W,N1), t1 (N,H,W,N2)).
    proj=(pk_d, scale_d, shift_d): `skip` is the block's INPUT (N,H,W,K1) and the projection shortcut conv_d + bn_d is computed
    inside the launch (tlxmi_bottleneck_seam_proj) instead of being read from a stored map."""
    need_gpu(t2, "input")
    N, H, W, ld = t2.shape
    if skip.shape[:3] != t2.shape[:3] or skip.dtype != t2.dtype or not skip.is_contiguous() or not t2.is_contiguous():
        raise RuntimeError("bottleneck_seam: t2 / skip must be dense NHWC maps of one dtype and extent")
    y = torch.empty((N, H, W, pk3.Cout), dtype=t2.dtype, device=t2.device)
    t1 = torch.empty((N, H, W, pk1.Cout), dtype=t2.dtype, device=t2.device)
    rows = N * H * W
    d = _lib.SeamDesc(dtype=dt_code(t2.dtype), rows=rows, K1=pk3.Cin, N1=pk3.Cout, N2=pk1.Cout, t2_ld=ld, skip_ld=skip.shape[-1],
                      y_ld=pk3.Cout, t1_ld=pk1.Cout, act=ACT_RELU)
    if proj is None:
        name = "tlxmi_bottleneck_seam"
        args = (C.byref(d), _p(t2), _p(pk3.buf), _p(scale3), _p(shift3), _p(skip), _p(y), _p(pk1.buf), _p(scale1), _p(shift1), _p(t1), _stream())
    else:
        name = "tlxmi_bottleneck_seam_proj"
        pkd, sd, hd = proj
        args = (C.byref(d), _p(t2), _p(pk3.buf), _p(scale3), _p(shift3), _p(skip), _p(pkd.buf), _p(sd), _p(hd), _p(y), _p(pk1.buf), _p(scale1),
                _p(shift1), _p(t1), _stream())
    if _probe is None:
        _lib.call(name, *args)
        return y, t1
    es = t2.element_size()
    skip_ch = pk3.Cout if proj is None else proj[0].Cin
    alg_bytes = (rows * (pk3.Cin + skip_ch + pk3.Cout + pk1.Cout) + pk3.Cout * pk3.Cin + pk1.Cout * pk1.Cin) * es
    flops = 2 * rows * (pk3.Cout * pk3.Cin + pk1.Cout * pk1.Cin + (0 if proj is None else pk3.Cout * proj[0].Cin))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.call(name, *args)
    e1.record()
    _probe.append((e0, e1, alg_bytes, flops, (N, H, W, pk3.Cin, pk3.Cout, pk1.Cout, "seam" if proj is None else "seam+proj", True)))
    return y, t1


def group_conv2d(x, pk, stride=1, padding=0, dilation=1, scale=None, shift=None, res=None, act=ACT_NONE,
                 act_param=0.0, res_after_act=False):
    """Grouped convolution (+ folded BatchNorm / bias, activation, residual): x (N,H,W,Cin) NHWC with exactly the
    filter's input channels -> (N,Ho,Wo,Cout).  pk: PackedGroupFilter."""
    need_gpu(x, "input")
    N, H, W, ld = x.shape
    if x.dtype != pk.dtype:
        raise RuntimeError(f"group_conv2d: input dtype {x.dtype} != packed filter dtype {pk.dtype}")
    if ld != pk.Cin:
        raise RuntimeError(f"group_conv2d: input has {ld} channels, filter expects {pk.Cin}")
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    Ho = (H + 2 * ph - dh * (pk.R - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dw * (pk.S - 1) - 1) // sw + 1
    if Ho <= 0 or Wo <= 0:
        raise RuntimeError(f"group_conv2d: empty output {Ho}x{Wo} for input {H}x{W}")
    out = torch.empty((N, Ho, Wo, pk.Cout), dtype=x.dtype, device=x.device)
    if res is not None and res.dtype != x.dtype:
        raise RuntimeError("group_conv2d: residual dtype mismatch")
    d = _lib.ConvDesc(dtype=dt_code(x.dtype), N=N, H=H, W=W, C=pk.Cin, Cout=pk.Cout, R=pk.R, S=pk.S,
                      stride_h=sh, stride_w=sw, pad_h=ph, pad_w=pw, dil_h=dh, dil_w=dw, Ho=Ho, Wo=Wo,
                      x_ld=ld, y_ld=pk.Cout, res_ld=(res.shape[-1] if res is not None else 0),
                      y_nstride=0, res_nstride=0, act=act, act_param=float(act_param),
                      flags=(EPI_RES_AFTER_ACT if res_after_act else 0) | plan_flags())
    args = (C.byref(d), pk.groups, _p(x), _p(pk.buf), _p(scale), _p(shift), _p(res), _p(out), _stream())
    if _probe is None:
        _lib.call("tlxmi_group_conv2d", *args)
        return out
    es = x.element_size()
    M = N * Ho * Wo
    cg = pk.Cin // pk.groups
    alg_bytes = (N * H * W * pk.Cin + M * pk.Cout * (2 if res is not None else 1) + pk.Cout * cg * pk.R * pk.S) * es
    flops = 2 * M * pk.Cout * cg * pk.R * pk.S
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.call("tlxmi_group_conv2d", *args)
    e1.record()
    _probe.append((e0, e1, alg_bytes, flops, (N, H, W, pk.Cin, pk.Cout, pk.R, sh, f"g{pk.groups}")))
    return out


def linear(x, pk, bias=None, res=None, act=ACT_NONE, out=None):
    """x (..., K) -> (..., Cout): the H=W=1 case of the implicit GEMM, bias as epilogue shift."""
    need_gpu(x, "input")
    shp = x.shape
    if not x.is_contiguous():
        x = x.contiguous()
    rows = x.numel() // shp[-1]
    x4 = x.view(rows, 1, 1, shp[-1])
    r4 = None
    if res is not None:
        if not res.is_contiguous():
            res = res.contiguous()
        r4 = res.view(rows, 1, 1, pk.Cout)
    o4 = None
    if out is not None:
        o4 = out.view(rows, 1, 1, pk.Cout)
    if x.dtype != pk.dtype:
        raise RuntimeError(f"linear: input dtype {x.dtype} != packed filter dtype {pk.dtype}")
    tail = _linear_tail(rows, shp[-1], pk, x) if (_probe is None and pk.R == 1 and pk.S == 1) else None
    if tail is not None:
        # Whole rounds of 256 x 256 tiles on the persistent kernel, the rows of the short last round as K slices side by side +
        # the deterministic slice-order reduction (bias, residual, activation there): the balanced tail WITHOUT giving up the
        # tile's arithmetic intensity (a K slice of a tile reads as many operand bytes per FLOP as the whole tile)
        m_lo, slices = tail
        y = out.view(rows, pk.Cout) if out is not None else torch.empty((rows, pk.Cout), dtype=x.dtype, device=x.device)
        x2 = x.view(rows, shp[-1])
        r2 = res.view(rows, pk.Cout) if res is not None else None
        conv2d(x2[:m_lo].view(m_lo, 1, 1, shp[-1]), pk, shift=bias, res=(r2[:m_lo].view(m_lo, 1, 1, pk.Cout) if r2 is not None else None),
               act=act, out=y[:m_lo].view(m_lo, 1, 1, pk.Cout))
        hi = rows - m_lo
        part = torch.empty((slices, hi, pk.Cout), dtype=torch.float32, device=x.device)      # fp32 partial sums
        _lib.call("tlxmi_linear_splitk", dt_code(x.dtype), hi, shp[-1], pk.Cout, shp[-1], _p(x2[m_lo:]), _p(pk.buf), slices, _p(part),
                  None, _p(bias), _p(r2[m_lo:] if r2 is not None else None), pk.Cout if r2 is not None else 0, act, C.c_float(0.0), 0,
                  _p(y[m_lo:]), pk.Cout, _stream())
        return y.view(*shp[:-1], pk.Cout)
    splits = _linear_splits(rows, shp[-1], pk, x) if (out is None and _probe is None and pk.R == 1 and pk.S == 1) else 0
    if splits:
        # few rows, large filter (classifier heads): K slices side by side + a deterministic reduction (tlxmi_linear_splitk)
        part = torch.empty((splits, rows, pk.Cout), dtype=torch.float32, device=x.device)      # fp32 partial sums
        y = torch.empty((rows, pk.Cout), dtype=x.dtype, device=x.device)
        _lib.call("tlxmi_linear_splitk", dt_code(x.dtype), rows, shp[-1], pk.Cout, shp[-1], _p(x), _p(pk.buf), splits, _p(part),
                  None, _p(bias), _p(res), pk.Cout if res is not None else 0, act, C.c_float(0.0), 0, _p(y), pk.Cout, _stream())
        return y.view(*shp[:-1], pk.Cout)
    y = conv2d(x4, pk, shift=bias, res=r4, act=act, out=o4)
    return y.view(*shp[:-1], pk.Cout)


def _linear_tail(rows, K, pk, x):
    """(rows of the whole rounds, K slices of the rest) for a Linear whose 256 x 256 tiles are r whole rounds on the CUs this
    launch is planned for plus a last round at most half full — ViT-B/16 proj / fc2 at batch 256: 2.32 rounds paid as 3, and
    batch 222 runs 6 % fewer images per second than batch 220 (tools/batch_quant.py) — else None.  The persistent kernel then
    runs exactly r rounds and the leftover rows go to tlxmi_linear_splitk.  OFF by default (set_option("tail_splitk", True) for the
    A/B): measured slower than the idle round it removes (the option table above); the in-kernel half-height tiles of
    gemm_stream.hip are what the product runs."""
    if not _options["tail_splitk"] or pk.Cin != K or pk.Cin_pad != K or pk.Cout < 256:
        return None
    es = x.element_size()
    kt = K * es // 128                      # K tiles of 128 bytes
    if (K * es) % 128 or kt < _TAIL_MIN_KTILES or (pk.Cout * es) % 16:
        return None
    idx = x.device.index if x.device.index is not None else torch.cuda.current_device()
    if idx not in _cus:
        _cus[idx] = torch.cuda.get_device_properties(idx).multi_processor_count
    cus = _cus[idx] // 2 if (plan_flags() & _lib.PLAN_SHARED_HALF) else _cus[idx]
    grid = max(8, cus & ~7)
    nt = (pk.Cout + 255) // 256
    tiles = ((rows + 255) // 256) * nt
    r, rem = divmod(tiles, grid)
    if r < 1 or rem == 0 or 2 * rem > grid:
        return None
    m_lo = (r * grid // nt) * 256
    hi = rows - m_lo
    if m_lo <= 0 or hi <= 0:
        return None
    t128 = ((hi + 127) // 128) * ((pk.Cout + 127) // 128)
    best = 0
    for s_ in range(2, 9):
        if kt % s_ or kt // s_ < 4:
            continue
        best = s_
        if t128 * s_ >= 2 * grid:
            break
    if not best or hi * pk.Cout * 4 * best >= (1 << 31):
        return None
    return m_lo, best


_TAIL_MIN_KTILES = 12      # (K >= 768 in fp16)


def _linear_splits(rows, K, pk, x):
    """Number of K slices for a Linear with few rows (0: run it as one GEMM).  set_option("splitk", False) turns the path off (A/B)."""
    if not _options["splitk"] or rows > 512 or pk.Cin != K or pk.Cin_pad != K:
        return 0
    es = x.element_size()
    if K * es < 4096 or pk.Cout * K * es < 3000000 or (pk.Cout * es) % 16:      # a filter of >= 3 MB with K >= 2048 (fp16): VGG / AlexNet
        # heads (round 2) and, round 4, ResNet's 2048 -> 1000 head (32 tiles of 64 x 64 per half batch, each walking all of K: 3.403 ->
        # 3.387 ms per ResNet-50 forward, tools/ab_tmp.py)
        return 0
    idx = x.device.index if x.device.index is not None else torch.cuda.current_device()
    if idx not in _cus:
        _cus[idx] = torch.cuda.get_device_properties(idx).multi_processor_count
    tiles = ((rows + 63) // 64) * ((pk.Cout + 63) // 64)
    kt = K * es // 128                      # K tiles of 128 bytes
    best = 0
    for s_ in range(2, 65):
        if kt % s_ or kt // s_ < 4:         # at least 4 K tiles per slice: the partial sums stay a small share of the traffic
            continue
        best = s_
        if tiles * s_ >= 3 * _cus[idx]:
            break
    if best and rows * pk.Cout * es * best >= (1 << 31):
        return 0
    return best


def mlp_seam_supported(rows, K, hidden, N, dtype):
    return bool(_options["mlp_seam"] and dtype == torch.float16 and rows >= 4096 and _lib.load().tlxmi_mlp_seam_supported(F16, int(K), int(hidden), int(N)))


def mlp_seam(x, pk1, b1, pk2, b2, res, out=None):
    """out = fc2(gelu(fc1(x) + b1)) + b2 + res in ONE launch (tlxmi_mlp_seam): x (..., K) fp16, pk1 / pk2 the packed fc1 / fc2 filters, res
    (..., N) the residual rows; out may be res (in place: a row is read before it is written by the lane that owns it)."""
    need_gpu(x, "input")
    shp = x.shape
    x = x if x.is_contiguous() else x.contiguous()
    K = shp[-1]
    rows = x.numel() // K
    N = pk2.Cout
    if not res.is_contiguous():
        raise RuntimeError("mlp_seam: the residual must be dense")
    y = out if out is not None else torch.empty((*shp[:-1], N), dtype=x.dtype, device=x.device)
    if _probe is not None:
        e0, e1 = _probe_pair()
    _lib.call("tlxmi_mlp_seam", dt_code(x.dtype), rows, K, pk1.Cout, N, _p(x), K, _p(pk1.buf), _p(b1), _p(pk2.buf), _p(b2), _p(res), N, _p(y), N, _stream())
    if _probe is not None:
        e1.record()
        _probe.append((e0, e1, (rows * (K + 2 * N) + 2 * pk1.Cout * K) * 2, 2 * rows * pk1.Cout * (K + N), (rows, 1, 1, K, N, pk1.Cout, "mlp seam", True)))
    return y


def linear_ln_supported(rows, K, Cout, dtype, act=ACT_NONE, with_res=False, producer=False):
    """Whether a Linear of this shape takes the folded-LayerNorm path (tlxmi_linear_stats with_res / tlxmi_linear_ln): fp16 on the
    persistent 256 x 256 GEMM kernel, rows enough to fill it (below ~2 k rows the tiled kernels of the dispatcher win)."""
    # tools/batch_table.py (round 5): inside a two-stream forward the other half's launches fill what a persistent GEMM with few tiles
    # leaves idle, so the fold pays from ~2 k rows (ViT-B/16 batch 64 = 2 x 32 images: +9 %); a forward on ONE stream at that size is
    # faster on the dispatcher's smaller tiles + LayerNorm launches (ViT-B/16 batch 32: -6 %, Swin-B batch 32: -7 %, batch 64: equal)
    min_rows = _options["lnfold_min_rows"] if in_halves() else max(_options["lnfold_min_rows"], _options["lnfold_min_rows_one_stream"])
    if not _options["lnfold"] or dtype != torch.float16 or rows < min_rows:
        return False
    return bool(_lib.load().tlxmi_linear_ln_supported(F16, int(rows), int(K), int(Cout), int(act), 1 if with_res else 2 if producer else 0))


class LinearLN:
    """A Linear with the LayerNorm in front of it folded in (include/tlxmi.h, tlxmi_linear_ln): packed W * gamma, c1 = row sums of the
    values as packed, c2 = bias + W @ beta."""

    def __init__(self, w_out_in, bias, gamma, beta, dtype):
        w = w_out_in.detach().float()
        g, b = gamma.detach().float().to(w.device), beta.detach().float().to(w.device)
        wg = w * g[None, :]
        self.pk = PackedFilter(wg.view(w.shape[0], w.shape[1], 1, 1).contiguous(), dtype)
        self.c1 = wg.to(dtype).double().sum(dim=1).float().contiguous()
        self.c2 = (w.double() @ b.double() + (bias.detach().double().to(w.device) if bias is not None else 0.0)).float().contiguous()
        self.K, self.Cout = w.shape[1], w.shape[0]


def _probe_pair():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    return e0, e1


def linear_stats(x, pk, bias=None, res=None, out=None):
    """y = x W^T + bias (+ res) as linear(), plus per row (sum, sum of squares) of y over every 256-channel tile column from the same
    epilogue: returns (y, partials (rows, 4, 2) fp32, pair p < ceil(Cout / 256) written) — the statistics of the LayerNorm that follows,
    without a pass over y; linear_ln takes them as they are."""
    need_gpu(x, "input")
    shp = x.shape
    if not x.is_contiguous():
        x = x.contiguous()
    K = shp[-1]
    rows = x.numel() // K
    if x.dtype != pk.dtype:
        raise RuntimeError(f"linear_stats: input dtype {x.dtype} != packed filter dtype {pk.dtype}")
    if res is not None and (not res.is_contiguous() or res.dtype != x.dtype):
        raise RuntimeError("linear_stats: the residual must be a dense tensor of the input's dtype")
    y = out if out is not None else torch.empty((*shp[:-1], pk.Cout), dtype=x.dtype, device=x.device)
    part = torch.empty((rows, 4, 2), dtype=torch.float32, device=x.device)      # pair p < ceil(Cout / 256) written
    if _probe is not None:
        e0, e1 = _probe_pair()
    _lib.call("tlxmi_linear_stats", dt_code(x.dtype), rows, K, pk.Cout, K, pk.Cout, _p(x), _p(pk.buf), _p(bias), _p(res),
              pk.Cout if res is not None else 0, _p(y), _p(part), plan_flags(), _stream())
    if _probe is not None:
        e1.record()
        es = x.element_size()
        _probe.append((e0, e1, (rows * K + rows * pk.Cout * (2 if res is not None else 1) + pk.Cout * K) * es + part.numel() * 4,
                       2 * rows * pk.Cout * K, (rows, 1, 1, K, pk.Cout, 1, 1, res is not None)))
    return y, part


def linear_ln(x, prep, part, eps, act=ACT_NONE):
    """act(Linear(LayerNorm(x))) on the RAW rows x (..., K): `part` = the (rows, 4, 2) pairs of (sum, sum of squares) the linear_stats
    launch that wrote x left (the first ceil(K / 256) of a row are read); mean / rstd of a row are formed inside the GEMM and applied in its epilogue."""
    need_gpu(x, "input")
    shp = x.shape
    if not x.is_contiguous():
        x = x.contiguous()
    rows = x.numel() // shp[-1]
    if tuple(part.shape) != (rows, 4, 2) or part.dtype != torch.float32 or not part.is_contiguous():
        raise RuntimeError(f"linear_ln: statistics of shape {tuple(part.shape)} for {rows} rows (expected {(rows, 4, 2)} fp32)")
    y = torch.empty((*shp[:-1], prep.Cout), dtype=x.dtype, device=x.device)
    if _probe is not None:
        e0, e1 = _probe_pair()
    _lib.call("tlxmi_linear_ln", dt_code(x.dtype), rows, prep.K, prep.Cout, prep.K, prep.Cout, _p(x), _p(prep.pk.buf), _p(prep.c1), _p(prep.c2),
              _p(part), C.c_float(float(eps)), act, _p(y), plan_flags(), _stream())
    if _probe is not None:
        e1.record()
        es = x.element_size()
        _probe.append((e0, e1, (rows * prep.K + rows * prep.Cout + prep.Cout * prep.K) * es + part.numel() * 4,
                       2 * rows * prep.Cout * prep.K, (rows, 1, 1, prep.K, prep.Cout, 1, 1, False)))
    return y


def dwconv2d(x, w_rsc, stride=1, padding=0, dilation=1, scale=None, shift=None, act=ACT_NONE, act_param=0.0, out_hw=None):
    need_gpu(x, "input")
    N, H, W, Cc = x.shape
    R, S, Cw = w_rsc.shape
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    Ho = (H + 2 * ph - dh * (R - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dw * (S - 1) - 1) // sw + 1
    if out_hw is not None:      # one-sided end padding ('SAME' at stride 2): the last windows read zeros past the edge
        Ho, Wo = int(out_hw[0]), int(out_hw[1])
    y = torch.empty((N, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
    d = _lib.DwConvDesc(dtype=dt_code(x.dtype), N=N, H=H, W=W, C=Cw, R=R, S=S, stride_h=sh, stride_w=sw, pad_h=ph,
                        pad_w=pw, dil_h=dh, dil_w=dw, Ho=Ho, Wo=Wo, x_ld=Cc, y_ld=Cc, act=act,
                        act_param=float(act_param))
    _lib.call("tlxmi_dwconv2d", C.byref(d), _p(x), _p(w_rsc), _p(scale), _p(shift), _p(y), _stream())
    return y


# ---------------------------------------------------------------------------------------------
# pooling / elementwise / norm
# ---------------------------------------------------------------------------------------------
def maxpool2d(x, kernel, stride, padding):
    need_gpu(x, "input")
    N, H, W, Cc = x.shape
    R, S = _pair(kernel)
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    Ho = (H + 2 * ph - R) // sh + 1
    Wo = (W + 2 * pw - S) // sw + 1
    y = torch.empty((N, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
    _lib.call("tlxmi_maxpool2d", _p(x), _p(y), dt_code(x.dtype), N, H, W, Cc, Cc, Cc, R, S, sh, sw, ph, pw, Ho, Wo,
              _stream())
    return y


def avgpool2d(x, kernel, stride, padding):
    """nn.AvgPool2d on an NHWC map; zero padding counts in the divisor."""
    need_gpu(x, "input")
    N, H, W, Cc = x.shape
    R, S = _pair(kernel)
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    Ho = (H + 2 * ph - R) // sh + 1
    Wo = (W + 2 * pw - S) // sw + 1
    y = torch.empty((N, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
    _lib.call("tlxmi_avgpool2d", _p(x), _p(y), dt_code(x.dtype), N, H, W, Cc, Cc, Cc, R, S, sh, sw, ph, pw, Ho, Wo, _stream())
    return y


def radix_gap(x, radix):
    """(N,H,W,radix*C) -> (N,C): mean over pixels of the sum of the radix splits (resnest.py:150-155)."""
    need_gpu(x, "input")
    N, H, W, RC = x.shape
    Cc = RC // radix
    g = torch.empty((N, Cc), dtype=x.dtype, device=x.device)
    _lib.call("tlxmi_radix_gap", _p(x), _p(g), dt_code(x.dtype), N, H * W, Cc, radix, RC, Cc, _stream())
    return g


def split_attention(x, logit, radix, cardinality):
    """x (N,H,W,radix*C), logit (N,radix*C) -> (N,H,W,C): rSoftmax over the radix axis (sigmoid for radix 1) and the
    weighted sum of the splits (resnest.py:53-82, 158-165)."""
    need_gpu(x, "input")
    N, H, W, RC = x.shape
    Cc = RC // radix
    y = torch.empty((N, H, W, Cc), dtype=x.dtype, device=x.device)
    ws = torch.empty((N, RC), dtype=torch.float32, device=x.device)      # attention weights, split order
    _lib.call("tlxmi_split_attention", _p(x), _p(logit), _p(ws), _p(y), dt_code(x.dtype), N, H * W, Cc, radix, cardinality,
              RC, logit.shape[-1], Cc, _stream())
    return y


def global_avgpool(x):
    """(N,H,W,C) or (N,L,C) -> (N,C) mean over the middle axes."""
    need_gpu(x, "input")
    N, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (N * Cc)
    y = torch.empty((N, Cc), dtype=x.dtype, device=x.device)
    _lib.call("tlxmi_global_avgpool", _p(x), _p(y), dt_code(x.dtype), N, HW, Cc, Cc, Cc, _stream())
    return y


def affine_act(x, scale=None, shift=None, res=None, act=ACT_NONE, act_param=0.0, res_after_act=False, out=None):
    """y = act(x*scale[c] + shift[c] (+res)) over the last axis."""
    need_gpu(x, "input")
    if not x.is_contiguous():
        x = x.contiguous()
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    y = out if out is not None else torch.empty_like(x)
    if res is not None and not res.is_contiguous():
        res = res.contiguous()
    _lib.call("tlxmi_affine_act", _p(x), _p(scale), _p(shift), _p(res), _p(y), dt_code(x.dtype), rows, Cc, Cc,
              Cc if res is not None else 0, Cc, act, float(act_param), EPI_RES_AFTER_ACT if res_after_act else 0,
              _stream())
    return y


def act_flat(x, act, act_param=0.0):
    """Elementwise activation of a tensor of ANY shape: runs over the flat storage, so the channel count need not be a
    multiple of the 16-byte vector width and the memory layout (contiguous / channels_last) is kept.  A buffer whose
    element count is not a whole number of chunks (or that is not dense) takes one padded copy."""
    need_gpu(x, "input")
    if x.dtype != _precision:
        x = x.to(_precision)
    v = vec(x.dtype)
    dense = x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last))
    if not dense:
        x = x.contiguous()
    n = x.numel()
    if n == 0:
        return x.clone()
    if n % v == 0 and x.data_ptr() % 16 == 0:
        y = torch.empty_like(x)          # preserve_format: same strides as the dense input
        _lib.call("tlxmi_affine_act", _p(x), None, None, None, _p(y), dt_code(x.dtype), n // v, v, v, 0, v, act,
                  float(act_param), 0, _stream())
        return y
    npad = (n + v - 1) // v * v
    buf = torch.zeros(npad, dtype=x.dtype, device=x.device)
    buf[:n] = x.reshape(-1) if x.is_contiguous() else x.permute(0, 2, 3, 1).reshape(-1)
    out = torch.empty_like(buf)
    _lib.call("tlxmi_affine_act", _p(buf), None, None, None, _p(out), dt_code(x.dtype), npad // v, v, v, 0, v, act,
              float(act_param), 0, _stream())
    if x.is_contiguous():
        return out[:n].view(x.shape)
    N, Cc, H, W = x.shape
    return out[:n].view(N, H, W, Cc).permute(0, 3, 1, 2)


def scale_channels(x, s):
    """x (N,H,W,C) * s (N,C) broadcast over pixels (Squeeze-Excitation gate)."""
    need_gpu(x, "input")
    N, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (N * Cc)
    y = torch.empty_like(x)
    _lib.call("tlxmi_scale_channels", _p(x), _p(s), _p(y), dt_code(x.dtype), N, HW, Cc, Cc, s.shape[-1], Cc, _stream())
    return y


def adaptive_avgpool2d(x, out_hw):
    """(N,H,W,C) -> (N,OH,OW,C), windows as nn.AdaptiveAvgPool2d (vgg.py:36-39)."""
    need_gpu(x, "input")
    N, H, W, Cc = x.shape
    OH, OW = out_hw
    if (OH, OW) == (H, W):
        return x
    y = torch.empty((N, OH, OW, Cc), dtype=x.dtype, device=x.device)
    _lib.call("tlxmi_adaptive_avgpool2d", _p(x), _p(y), dt_code(x.dtype), N, H, W, Cc, OH, OW, x.stride(2), Cc, _stream())
    return y


def layernorm(x, gamma, beta, eps):
    need_gpu(x, "input")
    if not x.is_contiguous():
        x = x.contiguous()
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    y = torch.empty_like(x)
    _lib.call("tlxmi_layernorm", _p(x), _p(gamma), _p(beta), _p(y), dt_code(x.dtype), rows, Cc, Cc, Cc, float(eps),
              _stream())
    return y


def layernorm_rows(x, rows, Cc, x_ld, gamma, beta, eps):
    """LayerNorm of `rows` rows of width Cc that sit x_ld elements apart in `x` -> dense (rows, Cc)."""
    need_gpu(x, "input")
    y = torch.empty((rows, Cc), dtype=x.dtype, device=x.device)
    _lib.call("tlxmi_layernorm", _p(x), _p(gamma), _p(beta), _p(y), dt_code(x.dtype), rows, Cc, x_ld, Cc,
              float(eps), _stream())
    return y


def broadcast_rows_into(vec_, out, rows, out_ld):
    """out[r*out_ld : r*out_ld + len(vec)] = vec for r < rows (cls-token row of every image)."""
    need_gpu(out, "output")
    Cc = vec_.numel()
    _lib.call("tlxmi_copy_channels", _p(vec_), _p(out), dt_code(out.dtype), rows, Cc, 0, out_ld, _stream())
    return out


def softmax(x, axis=-1):
    """tlx.ops.softmax / nn.Softmax on the device (tlxmi_softmax_rows): fp16 / fp32, any axis (another axis than the last is moved there
    and back: two layout copies)."""
    need_gpu(x, "input")
    if x.dtype not in (torch.float16, torch.float32):
        x = x.to(_precision)
    ax = axis % x.dim()
    if ax != x.dim() - 1:
        return softmax(x.movedim(ax, -1).contiguous(), -1).movedim(-1, ax)
    if not x.is_contiguous():
        x = x.contiguous()
    Cc = x.shape[-1]
    y = torch.empty_like(x)
    if x.numel():
        _lib.call("tlxmi_softmax_rows", _p(x), _p(y), dt_code(x.dtype), x.numel() // Cc, Cc, Cc, Cc, _stream())
    return y


def matmul(a, b, transpose_a=False, transpose_b=False):
    """tlx.matmul on libtlxmi (detr.py:1013 `tlx.matmul(q, k, transpose_b=True)`, vision_transformer.py:117-120): every (M, K) x (K, N)
    product of the broadcast batch is one tlxmi_conv2d launch (a Linear with the second operand packed as its filter) — a coverage path
    for reference-style layer-by-layer forwards (the engine's own models run fused attention kernels); fp16 or fp32, fp32 accumulation."""
    need_gpu(a, "matmul operand")
    need_gpu(b, "matmul operand")
    dt = a.dtype if a.dtype in (torch.float16, torch.float32) else _precision
    a, b = a.to(dt), b.to(dt)
    if a.dim() < 2 or b.dim() < 2:
        raise RuntimeError("matmul: operands of at least two dimensions are expected")
    if transpose_a:
        a = a.transpose(-1, -2)
    w = b if transpose_b else b.transpose(-1, -2)          # (..., N, K): the second operand as a Linear's [out][in] weight
    M, K, N = a.shape[-2], a.shape[-1], w.shape[-2]
    if w.shape[-1] != K:
        raise RuntimeError(f"matmul: inner dimensions {K} and {w.shape[-1]} differ")
    batch = torch.broadcast_shapes(a.shape[:-2], w.shape[:-2])
    a = a.expand(*batch, M, K).reshape(-1, M, K)
    w = w.expand(*batch, N, K).reshape(-1, N, K)
    v = vec(dt)
    Kp = (K + v - 1) // v * v
    if Kp != K:                                            # rows must be whole 16-byte chunks
        a = torch.nn.functional.pad(a, (0, Kp - K))
        w = torch.nn.functional.pad(w, (0, Kp - K))
    a = a.contiguous()
    wf = w.float().contiguous()
    out = torch.empty((a.shape[0], M, N), dtype=dt, device=a.device)
    for i in range(a.shape[0]):
        pk = PackedFilter(wf[i], dt)
        conv2d(a[i].view(M, 1, 1, Kp), pk, out=out[i].view(M, 1, 1, N), out_ld=N)
    return out.view(*batch, M, N)


def attention(qkv, heads, scale, bias=None, mask=None):
    """qkv (B, N, 3*heads*hd) packed as [3][heads][hd] -> (B, N, heads*hd)."""
    need_gpu(qkv, "qkv")
    if not qkv.is_contiguous():
        qkv = qkv.contiguous()
    B, N, C3 = qkv.shape
    hd = C3 // (3 * heads)
    out = torch.empty((B, N, heads * hd), dtype=qkv.dtype, device=qkv.device)
    nW = mask.shape[0] if mask is not None else 0
    d = _lib.AttnDesc(dtype=dt_code(qkv.dtype), B=B, Ntok=N, heads=heads, hd=hd, scale=float(scale), nW=nW)
    _lib.call("tlxmi_attention", C.byref(d), _p(qkv), _p(bias), _p(mask), _p(out), _stream())
    return out


def mha(q, k, v, heads, scale, mask=None, need_weights=False, batch_first=False):
    """General multi-head attention core (tlxmi_mha): q (Lq, B, D), k / v (Lk, B, D) sequence-first — or (B, L, D) with
    batch_first — any row / batch strides as long as the last axis is dense; mask: additive fp32 (Lq, Lk) or
    (B*heads, Lq, Lk).  Returns (out in q's layout, head-averaged weights (B, Lq, Lk) fp32 or None)."""
    for t in (q, k, v):
        need_gpu(t, "mha operand")
        if t.stride(-1) != 1:
            raise RuntimeError("mha: the feature axis must be dense")
    if not (q.dtype == k.dtype == v.dtype):
        raise RuntimeError("mha: q / k / v dtypes differ")
    bd, ld = (0, 1) if batch_first else (1, 0)
    B, Lq, Lk, D = q.shape[bd], q.shape[ld], k.shape[ld], q.shape[-1]
    hd = D // heads
    out = torch.empty(q.shape, dtype=q.dtype, device=q.device)
    avg = torch.empty((B, Lq, Lk), dtype=torch.float32, device=q.device) if need_weights else None
    mode = 0
    if mask is not None:
        mask = mask.to(device=q.device, dtype=torch.float32).contiguous()
        if tuple(mask.shape) == (Lq, Lk):
            mode = 1
        elif tuple(mask.shape) == (B * heads, Lq, Lk):
            mode = 2
        else:
            raise RuntimeError(f"mha: attn_mask shape {tuple(mask.shape)}; expected ({Lq}, {Lk}) or ({B * heads}, {Lq}, {Lk})")
    d = _lib.MhaDesc(dtype=dt_code(q.dtype), B=B, Lq=Lq, Lk=Lk, heads=heads, hd=hd, scale=float(scale), mask_mode=mode,
                     q_batch_stride=q.stride(bd), q_row_stride=q.stride(ld), k_batch_stride=k.stride(bd),
                     k_row_stride=k.stride(ld), v_batch_stride=v.stride(bd), v_row_stride=v.stride(ld),
                     out_batch_stride=out.stride(bd), out_row_stride=out.stride(ld))
    _lib.call("tlxmi_mha", C.byref(d), _p(q), _p(k), _p(v), _p(mask), _p(out), _p(avg), _stream())
    return out, avg


def yolo_box(heads_nhwc, anchors, num_classes, img_size, conf_thresh=0.005, downsample_ratio=32, clip_bbox=True, scale_x_y=1.0):
    """YOLOBox.__call__ (yolov3.py:558-579) on the device: `heads_nhwc` are the head maps (N,H,W,A*(5+C)) from the coarsest
    stride down, `anchors` the flat (w, h, w, h, ...) list per head (YOLOv3Head.mask_anchors), img_size (N,2) int32 (h, w).
    -> boxes (N, M, 4) fp32, scores (N, M, C) fp32 with M = sum A*H*W, heads appended in order."""
    N = heads_nhwc[0].shape[0]
    As = [len(a) // 2 for a in anchors]
    Ms = [A * h.shape[1] * h.shape[2] for A, h in zip(As, heads_nhwc)]
    Mtot = sum(Ms)
    dev = heads_nhwc[0].device
    boxes = torch.empty((N, Mtot, 4), dtype=torch.float32, device=dev)
    scores = torch.empty((N, Mtot, num_classes), dtype=torch.float32, device=dev)
    img = img_size.to(device=dev, dtype=torch.int32).contiguous()
    off = 0
    for i, (h, anc) in enumerate(zip(heads_nhwc, anchors)):
        need_gpu(h, "head map")
        if not h.is_contiguous() or h.shape[-1] != As[i] * (5 + num_classes):
            raise RuntimeError(f"yolo_box: head {i} must be a dense (N,H,W,{As[i] * (5 + num_classes)}) map")
        a = torch.tensor(anc, dtype=torch.float32, device=dev)
        _lib.call("tlxmi_yolo_box", _p(h), dt_code(h.dtype), N, As[i], num_classes, h.shape[1], h.shape[2], 1, _p(img), _p(a),
                  C.c_float(conf_thresh), int(downsample_ratio // 2 ** i), 1 if clip_bbox else 0, C.c_float(scale_x_y), _p(boxes), _p(scores),
                  Mtot, off, _stream())
        off += Ms[i]
    return boxes, scores


def yolo_iou_aware(head_nhwc, num_anchors, num_classes, factor):
    """YOLOv3Head's IoU-aware objectness (yolov3.py:355-376) on one NHWC head map (N,H,W,A*(6+C)) -> (N,H,W,A*(5+C))."""
    need_gpu(head_nhwc, "head map")
    N, H, W, Cc = head_nhwc.shape
    if Cc != num_anchors * (6 + num_classes) or not head_nhwc.is_contiguous():
        raise RuntimeError(f"yolo_iou_aware: a dense (N,H,W,{num_anchors * (6 + num_classes)}) map is expected")
    y = torch.empty((N, H, W, num_anchors * (5 + num_classes)), dtype=head_nhwc.dtype, device=head_nhwc.device)
    _lib.call("tlxmi_yolo_iou_aware", _p(head_nhwc), _p(y), dt_code(head_nhwc.dtype), N * H * W, int(num_anchors), int(num_classes),
              C.c_float(factor), _stream())
    return y


def multiclass_nms(boxes, scores, score_threshold=0.05, nms_threshold=0.5, keep_top_k=100, return_index=False):
    """tlx_multiclass_nms (detection/utils/ops.py:255-329) on the device: boxes (N,M,4), scores (N,M,C) fp32 ->
    (detections (N, keep_top_k, 6) rows (class, score, x1, y1, x2, y2), counts (N,) int32); return_index (yolov3.py:70-78, the
    for_mot post-process): also keep_index (N, keep_top_k) int32, each row's box among the M of its image, -1 past the count."""
    need_gpu(boxes, "boxes")
    boxes, scores = boxes.float().contiguous(), scores.float().contiguous()
    N, M, Cc = scores.shape
    ws = torch.empty(_lib.load().tlxmi_multiclass_nms_workspace_bytes(N, M), dtype=torch.uint8, device=boxes.device)
    det = torch.empty((N, keep_top_k, 6), dtype=torch.float32, device=boxes.device)
    cnt = torch.empty((N,), dtype=torch.int32, device=boxes.device)
    if return_index:
        idx = torch.empty((N, keep_top_k), dtype=torch.int32, device=boxes.device)
        _lib.call("tlxmi_multiclass_nms_index", _p(boxes), _p(scores), N, M, Cc, C.c_float(score_threshold), C.c_float(nms_threshold),
                  int(keep_top_k), _p(ws), _p(det), _p(cnt), _p(idx), _stream())
        return det, cnt, idx
    _lib.call("tlxmi_multiclass_nms", _p(boxes), _p(scores), N, M, Cc, C.c_float(score_threshold), C.c_float(nms_threshold), int(keep_top_k),
              _p(ws), _p(det), _p(cnt), _stream())
    return det, cnt


def attention_table(bias, mask, N):
    """bias (heads, N, N) + mask (nW, N, N) or None -> the pre-summed, padded table of tlxmi_attention_comb:
    (max(nW,1), heads, NP, NP) fp32, NP = 32 * ceil(N / 32).  Built once per layer (the reference adds the two on
    every forward, swin_transformer.py:205-220)."""
    NP = (N + 31) // 32 * 32
    nW = mask.shape[0] if mask is not None else 1
    t = torch.zeros((nW, bias.shape[0], NP, NP), dtype=torch.float32, device=bias.device)
    t[:, :, :N, :N] = bias.float()[None]
    if mask is not None:
        t[:, :, :N, :N] += mask.float().to(bias.device)[:, None]
    return t.contiguous()


def attention_comb(qkv, heads, scale, table, nW):
    """attention() with a table from attention_table(); fp16, head dim 32 / 64 / 96, N <= 256."""
    need_gpu(qkv, "input")
    B, N, C3 = qkv.shape
    hd = C3 // (3 * heads)
    out = torch.empty((B, N, heads * hd), dtype=qkv.dtype, device=qkv.device)
    d = _lib.AttnDesc(dtype=dt_code(qkv.dtype), B=B, Ntok=N, heads=heads, hd=hd, scale=float(scale), nW=nW)
    _lib.call("tlxmi_attention_comb", C.byref(d), _p(qkv), _p(table), _p(out), _stream())
    return out


def attention_windows(qkv, heads, scale, table, nW, H, W, ws, shift):
    """Swin's windowed attention on IMAGE-order rows (tlxmi_attention_windows): qkv (B, H * W, 3 * heads * hd) -> (B, H * W, heads * hd); the
    cyclic shift and the window partition / reverse (swin_transformer.py:316-333) are row arithmetic inside the kernel.  table:
    attention_table(); nW: 0 or the windows per image (with the shift mask in the table)."""
    need_gpu(qkv, "qkv")
    B, L, C3 = qkv.shape
    hd = C3 // (3 * heads)
    out = torch.empty((B, L, heads * hd), dtype=qkv.dtype, device=qkv.device)
    wpi = (H // ws) * (W // ws)
    d = _lib.AttnDesc(dtype=dt_code(qkv.dtype), B=B * wpi, Ntok=ws * ws, heads=heads, hd=hd, scale=float(scale), nW=nW)
    _lib.call("tlxmi_attention_windows", C.byref(d), _p(qkv), _p(table), _p(out), H, W, ws, shift, _stream())
    return out


def window_partition(x, ws, shift):
    """(B,H,W,C) -> (B*nW, ws*ws, C) with the cyclic shift folded in."""
    need_gpu(x, "input")
    B, H, W, Cc = x.shape
    y = torch.empty((B * (H // ws) * (W // ws), ws * ws, Cc), dtype=x.dtype, device=x.device)
    _lib.call("tlxmi_window_partition", _p(x), _p(y), dt_code(x.dtype), B, H, W, Cc, ws, shift, _stream())
    return y


def window_reverse(win, B, H, W, ws, shift, res=None):
    need_gpu(win, "input")
    Cc = win.shape[-1]
    y = torch.empty((B, H, W, Cc), dtype=win.dtype, device=win.device)
    _lib.call("tlxmi_window_reverse", _p(win), _p(res), _p(y), dt_code(win.dtype), B, H, W, Cc, ws, shift, _stream())
    return y


def layernorm_window_partition(x, gamma, beta, eps, ws, shift):
    """(B,H,W,C) -> window_partition(roll(LayerNorm(x), -shift)) as (B*nW, ws*ws, C) in one pass."""
    need_gpu(x, "input")
    B, H, W, Cc = x.shape
    y = torch.empty((B * (H // ws) * (W // ws), ws * ws, Cc), dtype=x.dtype, device=x.device)
    _lib.call("tlxmi_layernorm_window_partition", _p(x), _p(_f32(gamma)), _p(_f32(beta)), _p(y), dt_code(x.dtype), B, H, W, Cc,
              ws, shift, C.c_float(eps), _stream())
    return y


def window_reverse_layernorm(win, res, gamma, beta, eps, ws, shift):
    """sum = res + roll(window_reverse(win), +shift); returns (sum, LayerNorm(sum)), both (B,H,W,C), in one pass."""
    need_gpu(win, "input")
    B, H, W, Cc = res.shape
    s = torch.empty_like(res)
    y = torch.empty_like(res)
    _lib.call("tlxmi_window_reverse_layernorm", _p(win), _p(res), _p(_f32(gamma)), _p(_f32(beta)), _p(s), _p(y),
              dt_code(win.dtype), B, H, W, Cc, ws, shift, C.c_float(eps), _stream())
    return s, y


def patch_merge_layernorm(x, gamma, beta, eps):
    """(B,H,W,C) -> LayerNorm over 4C of PatchMerging's 2 x 2 gather + concat, (B, H/2 * W/2, 4C), in one pass."""
    need_gpu(x, "input")
    B, H, W, Cc = x.shape
    y = torch.empty((B, (H // 2) * (W // 2), 4 * Cc), dtype=x.dtype, device=x.device)
    _lib.call("tlxmi_patch_merge_layernorm", _p(x), _p(_f32(gamma)), _p(_f32(beta)), _p(y), dt_code(x.dtype), B, H, W, Cc,
              C.c_float(eps), _stream())
    return y


def patch_merge_gather(x):
    need_gpu(x, "input")
    B, H, W, Cc = x.shape
    y = torch.empty((B, H // 2, W // 2, 4 * Cc), dtype=x.dtype, device=x.device)
    _lib.call("tlxmi_patch_merge_gather", _p(x), _p(y), dt_code(x.dtype), B, H, W, Cc, _stream())
    return y


def upsample2x_into(x, out, c_off):
    need_gpu(x, "input")
    N, H, W, Cc = x.shape
    _lib.call("tlxmi_upsample2x_nearest", _p(x), _p(out), dt_code(x.dtype), N, H, W, Cc, Cc, out.shape[-1], c_off,
              _stream())
    return out


def copy_channels_into(x, out, c_off):
    """out[..., c_off:c_off+C] = x   (tlx.concat along channels, one pass per source)."""
    need_gpu(x, "input")
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    es = x.element_size()
    dst = C.c_void_p(out.data_ptr() + c_off * es)
    _lib.call("tlxmi_copy_channels", _p(x), dst, dt_code(x.dtype), rows, Cc, Cc, out.shape[-1], _stream())
    return out


def argmax_lastdim(x):
    need_gpu(x, "input")
    if not x.is_contiguous():
        x = x.contiguous()
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    out = torch.empty(x.shape[:-1], dtype=torch.int64, device=x.device)
    _lib.call("tlxmi_argmax_lastdim", _p(x), dt_code(x.dtype), rows, Cc, Cc, _p(out), _stream())
    return out
