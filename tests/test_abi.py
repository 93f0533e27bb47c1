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
    assert ctypes.sizeof(_lib.Layer) == 32 + 14 * 8             # struct_size + 5 ints + alpha = 28 -> 32, 14 pointers
    assert ctypes.sizeof(_lib.LossDesc) == 8 + 5 * 8 + 8        # struct_size + nout, 5 pointers, ldc + ylog
    assert ctypes.sizeof(_lib.LogprobDesc) == 16 + 6 * 8 + 40 + 8 + 8 + 3 * 8 + 8 + 8
    for cls in (_lib.Gemm, _lib.Layer, _lib.LogprobDesc, _lib.LossDesc):
        assert cls._fields_[0][0] == "struct_size" and cls().struct_size == ctypes.sizeof(cls)
    arr = _lib.sized_array(_lib.Layer, 3)
    assert [a.struct_size for a in arr] == [ctypes.sizeof(_lib.Layer)] * 3


def test_a_descriptor_of_another_layout_is_refused():
    """ABI 11: every descriptor struct starts with ``struct_size``; an entry handed a struct whose size field is not this
    library's sizeof returns LINNA_ERR_INVALID with both numbers in the message instead of reading fields at wrong offsets
    (host-only entry here: linna_program_describe walks a linna_layer_t array; the GPU suite covers linna_logprob_create)."""
    from linna_amd import _lib, nn
    lib = _lib.load()
    model = nn.ChtoModelv2(33, 33, None)
    assert "WIDE" in nn.describe_program(model)[1]             # (the binding's own array carries the size: accepted)
    arr = (_lib.Layer * 2)()                                   # (ctypes arrays are zero-filled: struct_size 0)
    buf = ctypes.create_string_buffer(256)
    rc = lib.linna_program_describe(arr, 2, 33, 16, 0, buf, 256)
    msg = lib.linna_last_error().decode()
    assert rc == -1 and "struct_size is 0" in msg and str(ctypes.sizeof(_lib.Layer)) in msg, (rc, msg)
    arr = _lib.sized_array(_lib.Layer, 2)
    arr[1].struct_size = 136                                   # the ABI 10 size of the same struct
    rc = lib.linna_program_describe(arr, 2, 33, 16, 0, buf, 256)
    assert rc == -1 and "struct_size is 136" in lib.linna_last_error().decode()


def test_no_gpu_means_loud_failure():
    import torch
    from linna_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.LinnaHipError):
        _lib.ctx()


def test_serving_programs_are_planned_on_the_host_without_a_gpu():
    """linna_program_describe: the segment program of the whole-network kernel for the reference's network class
    (nn.py:59-133) -- on the 16-row serving engine the hidden h of the three residual blocks (1000 -> 16, 500 -> 32,
    250 -> 64) are SIDE segments of 4 / 2 / 1 k chunks and two steps, outside the weight stream, and so is the 33 -> 33 last
    layer (R4: one SIDE step instead of a three-step WIDE run; 104 steps instead of 121); the small-batch engines keep them
    as SPLIT / WIDE segments; the bench's plain MLP has none."""
    import torch  # noqa: F401
    from linna_amd import nn
    m = nn.ChtoModelv2(33, 33, None)
    n16, t16 = nn.describe_program(m, 16)
    n4, t4 = nn.describe_program(m, 4)
    assert n16 == n4 == 10 and t16.startswith("ok G 104 ") and t4.startswith("ok G 121 ")
    seg16, seg4 = t16.splitlines()[1:], t4.splitlines()[1:]
    assert seg16[1].startswith("SIDE steps 2 passes 1 ncg 1 kc 4") and seg16[3].startswith("SIDE steps 2 passes 1 ncg 1 kc 2")
    assert seg4[1].startswith("SPLIT steps 8") and seg4[3].startswith("SPLIT steps 4")
    assert seg16[5].startswith("SIDE steps 2 passes 1 ncg 1 kc 1")
    assert [ln.split()[0] for ln in seg16] == ["WIDE", "SIDE", "WIDE", "SIDE", "SPLIT", "SIDE", "SPLIT", "WIDE", "SPLIT", "SIDE"]
    assert seg16[9].startswith("SIDE steps 1 passes 1 ncg 1 kc 1") and seg4[9].startswith("WIDE steps 3")
    nd, td = nn.describe_program(nn.ChtoModelv2(40, 1000, None), 16, dense_nout=1000)
    assert nd == 11 and td.count("SIDE") == 3 and td.splitlines()[-1].startswith("WIDE steps 63 passes 2")    # the inverse covariance: last segment
    nm, tm = nn.describe_program(nn.MLP(33, 33, None), 16)
    assert nm == 10 and "SIDE" not in tm and " grad 1" in tm.splitlines()[0]                       # forward + backward half


def test_dense_factor_planning_without_a_gpu():
    """``linna_program_describe(dense_nout < -1)``: the dense log-likelihood segment in its factored form under the three
    ``linna_dense_tri`` modes -- ChtoModelv2(40,1000) on the 16-row engine: 416 steps with the full factor, 384 with the
    second column pass started at row 512, 356 with the balanced block assignment (66 steps per wave in that segment);
    widths outside (960, 1024] keep the short second pass."""
    import torch  # noqa: F401
    from linna_amd import nn, _lib
    lib = _lib.load()
    prev = lib.linna_dense_tri(-1)
    try:
        m = nn.ChtoModelv2(40, 1000, None)
        got = {}
        for mode in (0, 1, 2):
            lib.linna_dense_tri(mode)
            n, t = nn.describe_program(m, 16, dense_nout=-1000)
            got[mode] = (int(t.split()[2]), t.strip().splitlines()[n])
        assert [got[k][0] for k in (0, 1, 2)] == [416, 384, 356]
        assert got[0][1].startswith("WIDE steps 63 passes 2") and " zext 0 " in got[0][1]
        assert " zext 31 " in got[1][1] and " zext -1 " in got[2][1]
        for nout, z in ((700, " zext 12 "), (960, " zext 28 "), (961, " zext -1 "), (1024, " zext -1 ")):
            n, t = nn.describe_program(nn.MLP(10, nout, None, width=64, depth=1), 16, dense_nout=-nout)
            assert z in t.strip().splitlines()[n], (nout, t)
    finally:
        lib.linna_dense_tri(prev)


def test_no_kernel_of_the_library_uses_scratch():
    """Every kernel of liblinna_hip.so keeps its working set in registers: the AMDGPU metadata of the bundled gfx950 code
    objects says 0 bytes of private segment for all of them.  (A change of the weight ring's refill order once left the
    merged training launch with 2.3 KB of scratch per lane and the step 40 % slower -- results unchanged, so no parity test
    could see it.)"""
    import codeobj as _codeobj
    from linna_amd import _lib
    ks = _codeobj.kernels(_lib.LIB_PATH)
    names = [k["name"] for k in ks]
    assert len(ks) >= 70 and sum("net_stream_kernel" in n for n in names) >= 24 and any("gemm_group_update_kernel" in n for n in names)
    spilling = [(k["name"], k["scratch"]) for k in ks if k["scratch"]]
    assert not spilling, spilling
    assert max(k["vgpr"] for k in ks) <= 256


def test_slice_fusion_switch_without_a_gpu():
    """``linna_slice_fusion``: a process-wide mask, queried with -1, returns the previous value, refuses masks outside bits 0-2
    with LINNA_ERR_INVALID and a text (no launch, no GPU needed)."""
    from linna_amd import _lib
    lib = _lib.load()
    prev = lib.linna_slice_fusion(-1)
    assert 0 <= prev <= 7
    try:
        assert lib.linna_slice_fusion(0) == prev and lib.linna_slice_fusion(-1) == 0
        assert lib.linna_slice_fusion(5) == 0 and lib.linna_slice_fusion(-1) == 5
        rc = lib.linna_slice_fusion(8)
        assert rc < 0 and "linna_slice_fusion" in lib.linna_last_error().decode()
        assert lib.linna_slice_fusion(-1) == 5                 # (a refused call changes nothing)
    finally:
        lib.linna_slice_fusion(prev)


def test_gradient_program_planning_without_a_gpu():
    """``linna_program_describe(dense_nout = -1)``: the program of the one-launch gradient for ChtoModelv2(33,33) on the 16-row
    engine -- the forward segments with the hidden h of the residual blocks as SIDE segments (R4: they pay in this launch
    too since its gates are sign bits and its tables come through the kernel-argument segment), then the dX chain down to
    the 33 inputs with d/dh of every block as a one-step SIDE segment: 107 + 112 steps in one weight stream."""
    import torch
    from linna_amd import nn
    n, txt = nn.describe_program(nn.ChtoModelv2(33, 33, None), 16, -1)
    lines = txt.strip().splitlines()
    assert n == 20 and lines[0].startswith("ok G 107 Gstride 219 nseg_f 10")
    kinds = [ln.split()[0] for ln in lines[1:21]]
    assert kinds[:10] == ["WIDE", "SIDE", "WIDE", "SIDE", "SPLIT", "SIDE", "SPLIT", "WIDE", "SPLIT", "WIDE"]
    assert kinds[10:] == ["WIDE", "WIDE", "SPLIT", "SIDE", "SPLIT", "SIDE", "WIDE", "SIDE", "WIDE", "SPLIT"]
    assert all(ln.startswith("SIDE steps 1 ") for ln in (lines[14], lines[16], lines[18]))          # d/dh: 125 -> 64, 250 -> 32, 500 -> 16
    assert lines[-2].startswith("SPLIT steps 8") and lines[-2].endswith("N 33")          # d lnP / d x of the 1000-wide first layer
    assert sum(int(ln.split()[2]) * int(ln.split()[4]) for ln in lines[1:11] if not ln.startswith("SIDE")) == 107
    # the signs of the forward activations (the backward's gates) are a bit matrix in LDS: 2688 columns x 16 rows fit
    assert lines[-1].startswith("lds ") and "2688 sign-bit columns" in lines[-1] and lines[-1].endswith("one launch")
    assert int(lines[-1].split()[1]) <= 160 * 1024


def test_no_cpp_exception_crosses_the_c_boundary():
    """include/linna_hip.h: "No C++ exception crosses the boundary".  The host side of the library plans programs with
    std::vector / std::string / std::unordered_map; every extern "C" entry is a function-try-block that turns an exception
    into LINNA_ERR_INTERNAL + a text.  linna_debug_raise throws inside such an entry: the process survives, the code and the
    text come back, and the next call works."""
    from linna_amd import _lib
    lib = _lib.load()
    for kind, word in ((1, "bad_alloc"), (2, "vector"), (3, "unknown type"), (4, "at")):
        rc = lib.linna_debug_raise(kind)
        assert rc == _lib.ERR_INTERNAL, (kind, rc)
        text = lib.linna_last_error().decode()
        assert "C++ exception at the C boundary" in text and word in text, text
        with pytest.raises(_lib.LinnaHipError, match="C\\+\\+ exception"):
            _lib.check(rc)
    assert lib.linna_debug_raise(0) == 0
    assert lib.linna_abi_version() == _lib.ABI_VERSION


def test_every_entry_definition_is_a_function_try_block():
    """Source-level: each function the header declares (bar the two that cannot throw: version, error text) is defined
    as `... ) try {` and closed by LINNA_CATCH_INT / LINNA_CATCH_SIZE in api.hip / comm.hip."""
    csrc = os.path.join(ROOT, "linna_amd", "csrc")
    src = open(os.path.join(csrc, "api.hip")).read() + open(os.path.join(csrc, "comm.hip")).read()
    guarded = set()
    for m in re.finditer(r"^(?:int|size_t) (linna_\w+)\(([^{;]*)\) try \{", src, flags=re.M):
        guarded.add(m.group(1))
    decl = set(header_functions()) - {"linna_abi_version", "linna_last_error"}
    assert decl <= guarded, sorted(decl - guarded)
    assert src.count("LINNA_CATCH_INT") + src.count("LINNA_CATCH_SIZE") >= len(decl)
