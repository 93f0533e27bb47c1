"""CPU: the C-ABI library loads, exports every symbol include/linna_hip.h declares, and the
ctypes signatures agree with the header's parameter counts."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "linna_hip.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|size_t|const char\*)\s+(linna_\w+)\s*\(([^;{}]*)\)\s*;", src):
        args = m.group(3).strip()
        n = 0 if args in ("", "void") else len(args.split(","))
        out[m.group(2)] = (m.group(1), n)
    return out


def test_library_is_built():
    from linna_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"


def test_every_declared_symbol_is_exported_and_bound():
    from linna_amd import _lib
    decl = header_functions()
    assert len(decl) >= 40
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in decl:
        assert hasattr(lib, name), "header declares %s but the library does not export it" % name
    assert set(decl) == set(_lib.EXPORTED), set(decl) ^ set(_lib.EXPORTED)
    for name, (ret, nargs) in decl.items():
        res, args = _lib._SIGNATURES[name]
        assert len(args) == nargs, "%s: header has %d parameters, binding %d" % (name, nargs, len(args))
        if ret == "size_t":
            assert res is ctypes.c_size_t
        elif ret == "const char*":
            assert res is ctypes.c_char_p
        else:
            assert res is ctypes.c_int


def test_load_checks_abi_version_without_gpu():
    from linna_amd import _lib
    lib = _lib.load()
    assert lib.linna_abi_version() == _lib.ABI_VERSION
    assert lib.linna_gemm_dot_slots(4096, 33) >= 1


def test_struct_layouts_match_header_sizes():
    """Sizes computed from the header's field lists (LP64)."""
    from linna_amd import _lib
    assert ctypes.sizeof(_lib.GemmPair) == 40
    assert ctypes.sizeof(_lib.ColMap) == 40
    assert ctypes.sizeof(_lib.Layer) == 24 + 14 * 8
    assert ctypes.sizeof(_lib.LossDesc) == 8 + 5 * 8 + 8


def test_no_gpu_means_loud_failure():
    import torch
    from linna_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.LinnaHipError):
        _lib.ctx()
