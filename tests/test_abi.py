"""CPU-side checks of the C-ABI boundary: the library loads and exports every entry point that
include/tlxmi.h declares, and the ctypes mirrors of the POD descriptors have the C layout."""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "tlxmi.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tlxmi_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from tlxcv_amd import _lib
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"libtlxmi.so does not export {n}"
    # and the ctypes table covers exactly the header
    assert sorted(_lib.ALL_SYMBOLS) == names
    # ... and the library exports nothing under the prefix that the header does not declare (a removed entry point stays removed)
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = sorted(l.split()[-1] for l in out.splitlines() if l.split()[-1].startswith("tlxmi_") and l.split()[-2] in "TW")
    assert exported == names


def test_version_and_error_string():
    from tlxcv_amd import _lib
    lib = _lib.load()
    assert lib.tlxmi_version() == 101
    assert isinstance(lib.tlxmi_last_error(), bytes)


def test_descriptor_layout_matches_c(tmp_path):
    from tlxcv_amd import _lib
    prog = tmp_path / "sz.c"
    prog.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "tlxmi.h"\n'
        'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(tlxmi_conv2d_desc), sizeof(tlxmi_dwconv2d_desc),'
        ' sizeof(tlxmi_attn_desc), offsetof(tlxmi_conv2d_desc, act_param), offsetof(tlxmi_conv2d_desc, flags),'
        ' offsetof(tlxmi_attn_desc, scale), sizeof(tlxmi_mha_desc), offsetof(tlxmi_mha_desc, q_batch_stride),'
        ' offsetof(tlxmi_mha_desc, out_row_stride));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), str(prog), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).split()
    got = [int(x) for x in out]
    want = [ctypes.sizeof(_lib.ConvDesc), ctypes.sizeof(_lib.DwConvDesc), ctypes.sizeof(_lib.AttnDesc),
            _lib.ConvDesc.act_param.offset, _lib.ConvDesc.flags.offset, _lib.AttnDesc.scale.offset,
            ctypes.sizeof(_lib.MhaDesc), _lib.MhaDesc.q_batch_stride.offset, _lib.MhaDesc.out_row_stride.offset]
    assert got == want


def test_bad_arguments_are_rejected_without_a_gpu():
    """Argument validation happens before any HIP call, so it is testable on CPU."""
    from tlxcv_amd import _lib
    lib = _lib.load()
    d = _lib.ConvDesc(dtype=7)
    rc = lib.tlxmi_conv2d(ctypes.byref(d), None, None, None, None, None, None, None)
    assert rc == -1 and b"null" in lib.tlxmi_last_error()
    assert lib.tlxmi_packed_filter_bytes(64, 3, 7, 7, 0) == 128 * 448 * 2   # Cin 3->8, K 392->448 halves (7 x 128 B)
    assert lib.tlxmi_packed_filter_bytes(64, 3, 7, 7, 1) == 128 * 224 * 4   # Cin 3->4, K 196->224 floats
    assert lib.tlxmi_packed_filter_bytes(0, 3, 7, 7, 0) == 0


def test_round4_entry_points_reject_bad_arguments_without_a_gpu():
    """tlxmi_patchify / tlxmi_patch_embed4 / tlxmi_patch_merge_layernorm validate before any HIP call: null and misaligned buffers,
    sizes the kernels' index arithmetic does not cover, dtypes — each with its own error code and message."""
    from tlxcv_amd import _lib
    lib = _lib.load()
    buf = (ctypes.c_char * 4096)()
    base = ctypes.addressof(buf)
    base += (-base) % 16
    p, q = ctypes.c_void_p(base), ctypes.c_void_p(base + 2048)
    odd = ctypes.c_void_p(base + 8)
    F16, F32 = 0, 1
    # patchify(src, sdt, dst, ddt, N, C, H, W, ps, lead, stream)
    assert lib.tlxmi_patchify(None, F32, q, F16, 1, 3, 32, 32, 16, 1, None) == -1 and b"null" in lib.tlxmi_last_error()
    assert lib.tlxmi_patchify(p, F32, q, F16, 1, 3, 32, 32, 12, 1, None) == -1 and b"multiple of 8" in lib.tlxmi_last_error()
    assert lib.tlxmi_patchify(p, F32, q, F16, 1, 3, 30, 32, 16, 1, None) == -1
    assert lib.tlxmi_patchify(p, 7, q, F16, 1, 3, 32, 32, 16, 1, None) == -1 and b"dtype" in lib.tlxmi_last_error()
    assert lib.tlxmi_patchify(odd, F32, q, F16, 1, 3, 32, 32, 16, 1, None) == -3 and b"aligned" in lib.tlxmi_last_error()
    # patch_embed4(x, xdt, w, bias, gamma, beta, y, N, H, W, D, eps, stream)
    f = ctypes.c_float(1e-5)
    assert lib.tlxmi_patch_embed4(None, F32, p, None, None, None, q, 1, 8, 8, 128, f, None) == -1
    assert lib.tlxmi_patch_embed4(p, F32, p, None, None, None, q, 1, 8, 10, 128, f, None) == -1 and b"multiples of 4" in lib.tlxmi_last_error()
    assert lib.tlxmi_patch_embed4(p, F32, p, None, p, None, q, 1, 8, 8, 128, f, None) == -1        # gamma without beta
    assert lib.tlxmi_patch_embed4(p, F32, p, None, None, None, q, 1, 8, 8, 100, f, None) == -2 and b"D=100" in lib.tlxmi_last_error()
    assert lib.tlxmi_patch_embed4(p, F32, p, odd, None, None, q, 1, 8, 8, 128, f, None) == -1      # misaligned bias
    # patch_merge_layernorm(x, gamma, beta, y, dt, B, H, W, C, eps, stream)
    assert lib.tlxmi_patch_merge_layernorm(None, p, p, q, F16, 1, 4, 4, 16, f, None) == -1
    assert lib.tlxmi_patch_merge_layernorm(p, p, p, q, F16, 1, 5, 4, 16, f, None) == -1 and b"even" in lib.tlxmi_last_error()
    assert lib.tlxmi_patch_merge_layernorm(p, p, p, q, F16, 1, 4, 4, 12, f, None) == -3 and b"16-byte" in lib.tlxmi_last_error()
    assert lib.tlxmi_patch_merge_layernorm(p, p, p, q, 9, 1, 4, 4, 16, f, None) == -1


def test_round5_entry_points_reject_bad_arguments_without_a_gpu():
    """The folded-LayerNorm trio, the Mlp seam, softmax_rows and the index NMS validate before any HIP call: null and misaligned
    buffers, fp32, shapes outside the kernels, a partial-sum / row-table layout that does not match — each with its own code."""
    from tlxcv_amd import _lib
    lib = _lib.load()
    buf = (ctypes.c_char * 8192)()
    base = ctypes.addressof(buf)
    base += (-base) % 16
    p, q, r, t = (ctypes.c_void_p(base + 1024 * i) for i in range(4))
    odd8, odd4 = ctypes.c_void_p(base + 8), ctypes.c_void_p(base + 4)
    F16, F32, NONE, RELU, GELU = 0, 1, 0, 1, 6
    err = lambda: lib.tlxmi_last_error()
    # linear_stats(dtype, rows, K, Cout, x_ld, y_ld, x, w, bias, res, res_ld, y, partials, flags, stream)
    assert lib.tlxmi_linear_stats(F16, 4096, 768, 768, 768, 768, None, q, None, None, 0, r, t, 0, None) == -1 and b"null" in err()
    assert lib.tlxmi_linear_stats(F32, 4096, 768, 768, 768, 768, p, q, None, None, 0, r, t, 0, None) == -2 and b"fp16 only" in err()
    assert lib.tlxmi_linear_stats(F16, 4096, 768, 768, 760, 768, p, q, None, None, 0, r, t, 0, None) == -1 and b"extent" in err()
    assert lib.tlxmi_linear_stats(F16, 4096, 768, 100, 768, 104, p, q, None, None, 0, r, t, 0, None) == -2 and b"outside" in err()
    assert lib.tlxmi_linear_stats(F16, 4096, 768, 768, 768, 768, p, q, None, None, 0, r, None, 0, None) == -1 and b"partials" in err()
    assert lib.tlxmi_linear_stats(F16, 4096, 768, 768, 768, 768, p, q, None, None, 0, r, odd8, 0, None) == -1 and b"partials" in err()
    assert lib.tlxmi_linear_stats(F16, 4096, 768, 768, 768, 768, p, q, None, t, 760, r, t, 0, None) == -1 and b"residual" in err()
    assert lib.tlxmi_linear_stats(F16, 4096, 768, 768, 768, 768, p, odd8, None, None, 0, r, t, 0, None) == -3
    # linear_ln(dtype, rows, K, Cout, x_ld, y_ld, x, w, c1, c2, partials, eps, act, y, flags, stream)
    assert lib.tlxmi_linear_ln(F16, 4096, 768, 2304, 768, 2304, p, q, r, r, None, 1e-5, NONE, t, 0, None) == -1 and b"partials" in err()
    assert lib.tlxmi_linear_ln(F16, 4096, 768, 2304, 768, 2304, p, q, r, r, odd8, 1e-5, NONE, t, 0, None) == -1 and b"partials" in err()
    assert lib.tlxmi_linear_ln(F16, 4096, 768, 2304, 768, 2304, p, q, r, r, r, -1.0, NONE, t, 0, None) == -1 and b"eps" in err()
    assert lib.tlxmi_linear_ln(F16, 4096, 768, 2304, 768, 2304, p, q, r, r, r, 1e-5, RELU, t, 0, None) == -2 and b"activation" in err()
    assert lib.tlxmi_linear_ln(F32, 4096, 768, 2304, 768, 2304, p, q, r, r, r, 1e-5, GELU, t, 0, None) == -2
    assert lib.tlxmi_linear_ln(F16, 4096, 1536, 4608, 1536, 4608, p, q, r, r, r, 1e-5, NONE, t, 0, None) == -2 and b"planes" in err()
    assert lib.tlxmi_linear_ln_supported(F16, 50432, 1536, 4608, NONE, 0) == 0 and lib.tlxmi_linear_ln_supported(F16, 50432, 3072, 768, NONE, 1) == 1
    assert lib.tlxmi_linear_ln_supported(F16, 3136, 2048, 1024, NONE, 2) == 1 and lib.tlxmi_linear_ln_supported(F16, 3136, 2048, 1536, NONE, 2) == 0      # producers: Cout <= 1024
    assert lib.tlxmi_linear_stats(F16, 4096, 768, 1536, 768, 1536, p, q, None, None, 0, r, t, 0, None) == -2 and b"4 pairs" in err()
    assert lib.tlxmi_linear_ln_supported(F16, 50432, 768, 2304, GELU, 0) == 1 and lib.tlxmi_linear_ln_supported(F32, 50432, 768, 2304, NONE, 0) == 0
    assert lib.tlxmi_linear_ln_supported(F16, 50432, 768, 2304, RELU, 0) == 0 and lib.tlxmi_linear_ln_supported(F16, 50432, 768, 100, NONE, 0) == 0
    # mlp_seam(dtype, rows, K, hidden, N, x, x_ld, w1, b1, w2, b2, res, res_ld, out, out_ld, stream)
    assert lib.tlxmi_mlp_seam_supported(F16, 128, 512, 128) == 1 and lib.tlxmi_mlp_seam_supported(F16, 256, 1024, 256) == 0
    assert lib.tlxmi_mlp_seam_supported(F32, 128, 512, 128) == 0 and lib.tlxmi_mlp_seam_supported(F16, 128, 500, 128) == 0
    assert lib.tlxmi_mlp_seam(F16, 6272, 128, 512, 128, None, 128, q, None, r, None, t, 128, t, 128, None) == -1 and b"null" in err()
    assert lib.tlxmi_mlp_seam(F16, 6272, 256, 1024, 256, p, 256, q, None, r, None, t, 256, t, 256, None) == -2 and b"no kernel" in err()
    assert lib.tlxmi_mlp_seam(F16, 6272, 128, 512, 128, p, 120, q, None, r, None, t, 128, t, 128, None) == -1 and b"extent" in err()
    assert lib.tlxmi_mlp_seam(F16, 6272, 128, 512, 128, p, 128, q, None, r, None, odd8, 128, t, 128, None) == -3
    # softmax_rows(x, y, dt, rows, C, x_ld, y_ld, stream)
    assert lib.tlxmi_softmax_rows(None, q, F16, 8, 10, 10, 10, None) == -1
    assert lib.tlxmi_softmax_rows(p, q, 5, 8, 10, 10, 10, None) == -1 and b"dtype" in err()
    assert lib.tlxmi_softmax_rows(p, q, F16, 8, 10, 8, 10, None) == -1
    # attention_windows(desc, qkv, comb, out, H, W, ws, shift, stream)
    d = _lib.AttnDesc(dtype=F16, B=2 * 64, Ntok=49, heads=4, hd=32, scale=0.17, nW=64)
    assert lib.tlxmi_attention_windows(ctypes.byref(d), p, None, r, 56, 56, 7, 3, None) == -1 and b"null" in err()
    assert lib.tlxmi_attention_windows(ctypes.byref(d), p, q, r, 56, 55, 7, 3, None) == -1 and b"windows of 7" in err()
    assert lib.tlxmi_attention_windows(ctypes.byref(d), p, q, r, 56, 56, 7, 7, None) == -1                  # shift < ws
    d.nW = 16
    assert lib.tlxmi_attention_windows(ctypes.byref(d), p, q, r, 56, 56, 7, 3, None) == -1 and b"nW = 16" in err()
    d.nW, d.hd = 64, 48
    assert lib.tlxmi_attention_windows(ctypes.byref(d), p, q, r, 56, 56, 7, 3, None) == -2
    # patch_embed4_pos(x, xdt, w, bias, gamma, beta, pos, y, N, H, W, D, eps, stream)
    assert lib.tlxmi_patch_embed4_pos(p, F16, q, None, None, None, None, r, 1, 32, 32, 128, 1e-5, None) == -1 and b"position" in err()
    # multiclass_nms_index(boxes, scores, N, M, C, thr, nms_thr, keep_top_k, workspace, detections, counts, keep_index, stream)
    assert lib.tlxmi_multiclass_nms_index(p, q, 1, 100, 3, 0.1, 0.5, 10, r, t, t, None, None) == -1 and b"keep_index" in err()
    assert lib.tlxmi_multiclass_nms_index(p, q, 1, 100, 3, 0.1, 0.5, 0, r, t, t, t, None) == -1
    assert lib.tlxmi_multiclass_nms_index(p, q, 1, 70000, 3, 0.1, 0.5, 10, r, t, t, t, None) == -2


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from tlxcv_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_host_dispatch_survives_without_a_device():
    """The tile cost model / launch plumbing is host code: drive it with fake (aligned, never dereferenced on
    the host) pointers over the shapes of the three headline models.  Without a GPU the launch itself fails
    with TLXMI_ERR_LAUNCH (-4) — what must not happen is a host crash (division by zero in the model, ...)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("would launch on the device with fake pointers")
    from tlxcv_amd import _lib
    lib = _lib.load()
    shapes = [(256, 224, 224, 8, 64, 7, 2, 3), (256, 56, 56, 64, 256, 1, 1, 0), (256, 14, 14, 256, 256, 3, 1, 1),
              (50432, 1, 1, 768, 2304, 1, 1, 0), (50432, 1, 1, 3072, 768, 1, 1, 0), (256, 1, 1, 2048, 1000, 1, 1, 0),
              (6272, 1, 1, 128, 384, 1, 1, 0), (1, 5, 5, 8, 3, 3, 1, 1)]
    for dt in (0, 1):
        for (N, H, W, Cc, Co, k, s, p) in shapes:
            Ho = (H + 2 * p - k) // s + 1
            d = _lib.ConvDesc(dtype=dt, N=N, H=H, W=W, C=Cc, Cout=Co, R=k, S=k, stride_h=s, stride_w=s, pad_h=p,
                              pad_w=p, dil_h=1, dil_w=1, Ho=Ho, Wo=Ho if W > 1 else 1, x_ld=Cc, y_ld=Co, res_ld=Co,
                              act=1)
            for res in (None, ctypes.c_void_p(1 << 24)):
                rc = lib.tlxmi_conv2d(ctypes.byref(d), ctypes.c_void_p(1 << 20), ctypes.c_void_p(1 << 22), None, None,
                                      res, ctypes.c_void_p(1 << 26), None)
                assert rc in (0, -4), (rc, lib.tlxmi_last_error())


def test_product_library_reads_no_environment_knob():
    """the A/B knobs (TLXMI_TILE / HALO / TAIL / PP128 / STORE / DEBUG ...) exist only in the tuning flavour
    (libtlxmi_tune.so, -DTLXMI_TUNING).  The product library must not even import getenv."""
    from tlxcv_amd import _lib
    out = subprocess.check_output(["nm", "-D", "--undefined-only", _lib.LIB_PATH]).decode()
    assert "getenv" not in out
    if os.path.exists(_lib.TUNE_LIB_PATH):
        assert "getenv" in subprocess.check_output(["nm", "-D", "--undefined-only", _lib.TUNE_LIB_PATH]).decode()
        with _lib.tuning(TLXMI_TILE="3") as lib:
            assert os.environ["TLXMI_TILE"] == "3" and lib.tlxmi_version() == 101 and _lib.load() is lib
        assert "TLXMI_TILE" not in os.environ and _lib.load() is not lib


def test_product_loader_ignores_TLXMI_LIB_and_ships_no_tuning_only_kernel():
    """the product LOADER reads no environment variable either (a fresh interpreter with TLXMI_LIB pointing somewhere
    else still loads the in-tree libtlxmi.so), and the kernels the dispatcher never picks (gemm_w4.hip) are in the tuning flavour only."""
    from tlxcv_amd import _lib
    code = ("import os, tlxcv_amd._lib as L; lib = L.load(); "
            "assert L.LIB_PATH.endswith(os.path.join('tlxcv_amd', 'libtlxmi.so')), L.LIB_PATH; "
            "assert lib._name == L.LIB_PATH, lib._name; print('ok')")
    env = dict(os.environ, TLXMI_LIB="/nonexistent/libtlxmi_other.so")
    out = subprocess.check_output([sys.executable, "-c", code], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))).decode()
    assert out.strip().endswith("ok")
    syms = subprocess.check_output(["nm", "-C", _lib.LIB_PATH]).decode()
    assert "gemm_w4" not in syms
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gemm_w4" not in blob
    if os.path.exists(_lib.TUNE_LIB_PATH):
        assert "gemm_w4" in subprocess.check_output(["nm", "-C", _lib.TUNE_LIB_PATH]).decode()
