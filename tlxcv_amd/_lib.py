"""ctypes binding of libtlxmi.so (the C-ABI declared in include/tlxmi.h).

The product path has no CPU fallback: if the shared library is missing, or a call returns a
non-zero status, a RuntimeError is raised (the reference's convention for bad input is a plain
Python exception / assert, e.g. tlxcv/models/classification/vision_transformer.py:217-219).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtlxmi.so")   # the product loader reads no environment variable (two builds on one box: TLXMI_TUNE_LIB
                                                # of the tuning loader below, tools/ab_oldnew.sh)

F16, F32 = 0, 1
ACT_NONE, ACT_RELU, ACT_RELU6, ACT_LEAKY, ACT_HARDSWISH, ACT_HARDSIGMOID, ACT_GELU, ACT_SIGMOID, ACT_SILU = range(9)
EPI_RES_AFTER_ACT = 1
EPI_MAXPOOL_3S2P1 = 4
PLAN_SHARED_HALF = 0x100      # planning hints in the same flags word (include/tlxmi.h)
PLAN_SHARED_FULL = 0x200


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "dtype", "N", "H", "W", "C", "Cout", "R", "S", "stride_h", "stride_w", "pad_h", "pad_w",
        "dil_h", "dil_w", "Ho", "Wo", "x_ld", "y_ld", "res_ld", "y_nstride", "res_nstride", "act")] + [
        ("act_param", C.c_float), ("flags", C.c_uint32)]


class DwConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "dtype", "N", "H", "W", "C", "R", "S", "stride_h", "stride_w", "pad_h", "pad_w", "dil_h",
        "dil_w", "Ho", "Wo", "x_ld", "y_ld", "act")] + [("act_param", C.c_float)]


class AttnDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dtype", "B", "Ntok", "heads", "hd")] + [
        ("scale", C.c_float), ("nW", C.c_int32)]


class MhaDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dtype", "B", "Lq", "Lk", "heads", "hd")] + [
        ("scale", C.c_float), ("mask_mode", C.c_int32)] + [(n, C.c_int64) for n in (
            "q_batch_stride", "q_row_stride", "k_batch_stride", "k_row_stride", "v_batch_stride", "v_row_stride",
            "out_batch_stride", "out_row_stride")]


class SeamDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("rows", C.c_int64)] + [(n, C.c_int32) for n in (
        "K1", "N1", "N2", "t2_ld", "skip_ld", "y_ld", "t1_ld", "act")]


class PreprocDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("N", "H", "W", "C", "out_h", "out_w", "kh", "kw", "out_dtype", "layout", "fold_b",
                                         "cpad", "normalize")]


_vp, _i, _f, _u, _l = C.c_void_p, C.c_int, C.c_float, C.c_uint32, C.c_int64

# name -> argtypes; every function returns int status unless listed in _SPECIAL
PROTOTYPES = {
    "tlxmi_nchw_to_nhwc": [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "tlxmi_nchw_to_nhwc_s2d": [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "tlxmi_patchify": [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "tlxmi_patch_embed4": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp],
    "tlxmi_patch_embed4_pos": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp],
    "tlxmi_nhwc_to_nchw": [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "tlxmi_pack_filter": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "tlxmi_fold_bn": [_vp, _vp, _vp, _vp, _vp, _f, _i, _vp, _vp, _vp],
    "tlxmi_conv2d": [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "tlxmi_conv2d_splitk": [C.POINTER(ConvDesc), _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "tlxmi_bottleneck_seam": [C.POINTER(SeamDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "tlxmi_bottleneck_seam_proj": [C.POINTER(SeamDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "tlxmi_preprocess_u8": [C.POINTER(PreprocDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "tlxmi_preprocess_linear_u8": [C.POINTER(PreprocDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "tlxmi_yolo_iou_aware": [_vp, _vp, _i, _l, _i, _i, _f, _vp],
    "tlxmi_yolo_box": [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _f, _i, _i, _f, _vp, _vp, _i, _i, _vp],
    "tlxmi_multiclass_nms": [_vp, _vp, _i, _i, _i, _f, _f, _i, _vp, _vp, _vp, _vp],
    "tlxmi_multiclass_nms_index": [_vp, _vp, _i, _i, _i, _f, _f, _i, _vp, _vp, _vp, _vp, _vp],
    "tlxmi_linear_splitk": [_i, _l, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _f, _u, _vp, _i, _vp],
    "tlxmi_group_conv2d": [C.POINTER(ConvDesc), _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "tlxmi_pack_group_filter": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "tlxmi_dwconv2d": [C.POINTER(DwConvDesc), _vp, _vp, _vp, _vp, _vp, _vp],
    "tlxmi_maxpool2d": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "tlxmi_avgpool2d": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "tlxmi_radix_gap": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "tlxmi_split_attention": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "tlxmi_global_avgpool": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "tlxmi_adaptive_avgpool2d": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "tlxmi_affine_act": [_vp, _vp, _vp, _vp, _vp, _i, _l, _i, _i, _i, _i, _i, _f, _u, _vp],
    "tlxmi_scale_channels": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "tlxmi_layernorm": [_vp, _vp, _vp, _vp, _i, _l, _i, _i, _i, _f, _vp],
    "tlxmi_layernorm_window_partition": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp],
    "tlxmi_window_reverse_layernorm": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp],
    "tlxmi_mlp_seam": [_i, _l, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp],
    "tlxmi_linear_stats": [_i, _l, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _u, _vp],
    "tlxmi_softmax_rows": [_vp, _vp, _i, _l, _i, _l, _l, _vp],
    "tlxmi_linear_ln": [_i, _l, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _i, _vp, _u, _vp],
    "tlxmi_attention": [C.POINTER(AttnDesc), _vp, _vp, _vp, _vp, _vp],
    "tlxmi_attention_comb": [C.POINTER(AttnDesc), _vp, _vp, _vp, _vp],
    "tlxmi_attention_windows": [C.POINTER(AttnDesc), _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "tlxmi_mha": [C.POINTER(MhaDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "tlxmi_window_partition": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "tlxmi_window_reverse": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "tlxmi_patch_merge_gather": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "tlxmi_patch_merge_layernorm": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp],
    "tlxmi_upsample2x_nearest": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "tlxmi_copy_channels": [_vp, _vp, _i, _l, _i, _i, _i, _vp],
    "tlxmi_argmax_lastdim": [_vp, _i, _l, _i, _i, _vp, _vp],
}
_SPECIAL = {
    "tlxmi_version": ([], C.c_int),
    "tlxmi_last_error": ([], C.c_char_p),
    "tlxmi_device_count": ([], C.c_int),
    "tlxmi_packed_filter_bytes": ([_i, _i, _i, _i, _i], C.c_size_t),
    "tlxmi_packed_group_filter_bytes": ([_i, _i, _i, _i, _i, _i], C.c_size_t),
    "tlxmi_group_conv_chunks": ([_i, _i, _i, _i], C.c_int),
    "tlxmi_group_conv2d_small_supported": ([C.POINTER(ConvDesc), _i], C.c_int),
    "tlxmi_conv2d_maxpool_supported": ([C.POINTER(ConvDesc)], C.c_int),
    "tlxmi_conv2d_splitk_supported": ([C.POINTER(ConvDesc), _i], C.c_int),
    "tlxmi_preprocess_u8_workspace_bytes": ([C.POINTER(PreprocDesc)], C.c_size_t),
    "tlxmi_multiclass_nms_workspace_bytes": ([_i, _i], C.c_size_t),
    "tlxmi_bottleneck_seam_supported": ([_i, _i, _i, _i], C.c_int),
    "tlxmi_linear_ln_supported": ([_i, _l, _i, _i, _i, _i], C.c_int),
    "tlxmi_mlp_seam_supported": ([_i, _i, _i, _i], C.c_int),
}
ALL_SYMBOLS = sorted(list(PROTOTYPES) + list(_SPECIAL))

TUNE_LIB_PATH = os.environ.get("TLXMI_TUNE_LIB") or os.path.join(_HERE, "libtlxmi_tune.so")   # `make -C tlxcv_amd/csrc tune` (-DTLXMI_TUNING); TLXMI_TUNE_LIB: A/B two builds (tools/)

_lib = None        # the library call() goes through: the product build, or the tuning flavour inside `with tuning():`
_product = None
_tune = None


def _open(path):
    lib = C.CDLL(path)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    for name, (argtypes, restype) in _SPECIAL.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    return lib


def load():
    """Load libtlxmi.so (built by __graft_entry__.build() / `make -C tlxcv_amd/csrc`)."""
    global _lib, _product
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"tlxcv_amd: {LIB_PATH} is missing — the HIP engine has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C tlxcv_amd/csrc`). "
            "There is no CPU fallback.")
    _product = _open(LIB_PATH)
    _lib = _product
    return _lib


class tuning:
    """`with _lib.tuning(TLXMI_TILE="7"): ...` — run the enclosed calls on the TUNING flavour of the library
    (libtlxmi_tune.so: the same sources built with -DTLXMI_TUNING), which reads the A/B knobs (TLXMI_TILE, TLXMI_HALO,
    TLXMI_TAIL, TLXMI_PP128, TLXMI_STORE, TLXMI_DWSTRIP, TLXMI_PANEL_KB, TLXMI_DEBUG) from the environment on every call.
    The product library never does: outside this context no environment variable changes a kernel choice or a result.
    Used by tools/ab_*.py and by the tests that force a tile candidate the dispatcher would not pick at test sizes."""

    def __init__(self, **env):
        self.env = {k: str(v) for k, v in env.items()}

    def __enter__(self):
        global _lib, _tune
        load()
        if _tune is None:
            if not os.path.exists(TUNE_LIB_PATH):
                raise RuntimeError(f"tlxcv_amd: {TUNE_LIB_PATH} is missing — build it with `make -C tlxcv_amd/csrc tune`")
            _tune = _open(TUNE_LIB_PATH)
        self.saved = {k: os.environ.get(k) for k in self.env}
        os.environ.update(self.env)
        self.prev = _lib
        _lib = _tune
        return _tune

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev
        for k, v in self.saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        return False


def call(name, *args):
    """Invoke a status-returning entry point; non-zero status -> RuntimeError."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.tlxmi_last_error()
        raise RuntimeError(f"{name} failed ({rc}): {msg.decode() if msg else ''}")
