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


def test_version_and_error_string():
    from tlxcv_amd import _lib
    lib = _lib.load()
    assert lib.tlxmi_version() == 100
    assert isinstance(lib.tlxmi_last_error(), bytes)


def test_descriptor_layout_matches_c(tmp_path):
    from tlxcv_amd import _lib
    prog = tmp_path / "sz.c"
    prog.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "tlxmi.h"\n'
        'int main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(tlxmi_conv2d_desc), sizeof(tlxmi_dwconv2d_desc),'
        ' sizeof(tlxmi_attn_desc), offsetof(tlxmi_conv2d_desc, act_param), offsetof(tlxmi_conv2d_desc, flags),'
        ' offsetof(tlxmi_attn_desc, scale));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), str(prog), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).split()
    got = [int(x) for x in out]
    want = [ctypes.sizeof(_lib.ConvDesc), ctypes.sizeof(_lib.DwConvDesc), ctypes.sizeof(_lib.AttnDesc),
            _lib.ConvDesc.act_param.offset, _lib.ConvDesc.flags.offset, _lib.AttnDesc.scale.offset]
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


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from tlxcv_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()
