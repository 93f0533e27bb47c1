"""ctypes binding of liblinna_hip.so (C ABI in include/linna_hip.h).

PyTorch is used only as the owner of device memory and streams: every call passes
``tensor.data_ptr()`` and the raw ``hipStream_t`` of torch's current stream.  There is no
fallback: if the shared library is missing or the call fails, this module raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LINNA_LIB_PATH") or os.path.join(_HERE, "liblinna_hip.so")   # (LINNA_LIB_PATH: a diagnostic build, tools/ns_stamps.py)
ABI_VERSION = 11

c_float_p = C.c_void_p   # device pointers travel as void*
c_int_p = C.c_void_p


class _Sized(C.Structure):
    """A descriptor struct of include/linna_hip.h that starts with ``uint32_t struct_size`` (ABI 11): filled in on
    construction; elements of a ctypes ARRAY are not constructed -- use ``sized_array``."""

    def __init__(self, *a, **k):
        C.Structure.__init__(self, *a, **k)
        self.struct_size = C.sizeof(type(self))


def sized_array(cls, n):
    arr = (cls * n)()
    for i in range(n):
        arr[i].struct_size = C.sizeof(cls)
    return arr


class GemmPair(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("lda", C.c_int), ("ldb", C.c_int), ("K", C.c_int),
                ("alay", C.c_int), ("blay", C.c_int)]


class Gemm(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("p", GemmPair * 2), ("npairs", C.c_int), ("M", C.c_int), ("N", C.c_int),
                ("C", C.c_void_p), ("ldc", C.c_int), ("bias0", C.c_void_p), ("bias1", C.c_void_p),
                ("alpha0", C.c_float), ("R", C.c_void_p), ("ldr", C.c_int), ("relu", C.c_int),
                ("mask", C.c_void_p), ("ldmask", C.c_int), ("cscale", C.c_void_p), ("cshift", C.c_void_p),
                ("cexp", C.c_int), ("cpost", C.c_void_p), ("cshift2", C.c_void_p),
                ("dotwith", C.c_void_p), ("lddot", C.c_int), ("dot_partial", C.c_void_p), ("dot_slots", C.c_int),
                ("flags", C.c_int)]


class Layer(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("op", C.c_int), ("K", C.c_int), ("C", C.c_int), ("N", C.c_int), ("relu", C.c_int),
                ("alpha", C.c_float), ("W", C.c_void_p), ("b", C.c_void_p),
                ("W1", C.c_void_p), ("b1", C.c_void_p), ("W2", C.c_void_p), ("b2", C.c_void_p), ("Ws", C.c_void_p),
                ("gW", C.c_void_p), ("gb", C.c_void_p), ("gW1", C.c_void_p), ("gb1", C.c_void_p),
                ("gW2", C.c_void_p), ("gb2", C.c_void_p), ("gWs", C.c_void_p)]


class ColMap(C.Structure):
    _fields_ = [("cscale", C.c_void_p), ("cshift", C.c_void_p), ("cexp", C.c_int), ("cpost", C.c_void_p),
                ("cshift2", C.c_void_p)]


class LogprobDesc(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("nin", C.c_int), ("nout", C.c_int), ("is_flat", C.c_void_p), ("a1", C.c_void_p), ("a2", C.c_void_p),
                ("log10_flag", C.c_void_p), ("xmean", C.c_void_p), ("xstd", C.c_void_p), ("outmap", ColMap),
                ("S", C.c_void_p), ("lds", C.c_int), ("Ssym", C.c_void_p), ("w", C.c_void_p), ("gscale", C.c_void_p),
                ("temperature", C.c_float), ("Sfac", C.c_void_p)]


class LossDesc(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("nout", C.c_int), ("sigma", C.c_void_p), ("ymean", C.c_void_p), ("ystd", C.c_void_p),
                ("data_norm", C.c_void_p), ("Cinv", C.c_void_p), ("ldc", C.c_int), ("ylog", C.c_int)]


OP_LINEAR, OP_RESBLOCK, OP_INSKIP = 0, 1, 2
COMM_ID_BYTES = 128
LAY_K, LAY_MN = 0, 1

_V, _I, _F, _SZ, _U64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_uint64
_I64, _D = C.c_int64, C.c_double
_PV = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); mirrors include/linna_hip.h one to one
_SIGNATURES = {
    "linna_abi_version": (_I, []),
    "linna_last_error": (C.c_char_p, []),
    "linna_debug_raise": (_I, [_I]),
    "linna_ctx_create": (_I, [_I, _PV]),
    "linna_ctx_destroy": (_I, [_V]),
    "linna_stream_sync": (_I, [_V]),
    "linna_graph_begin": (_I, [_V]),
    "linna_graph_end": (_I, [_V, _PV]),
    "linna_graph_launch": (_I, [_V, _V]),
    "linna_graph_destroy": (_I, [_V]),
    "linna_event_create": (_I, [_PV]),
    "linna_event_record": (_I, [_V, _V]),
    "linna_event_elapsed_ms": (_I, [_V, _V, C.POINTER(C.c_float)]),
    "linna_event_destroy": (_I, [_V]),
    "linna_comm_unique_id": (_I, [_V]),
    "linna_comm_init": (_I, [_V, _I, _I, _V]),
    "linna_comm_destroy": (_I, [_V]),
    "linna_comm_info": (_I, [_V, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "linna_allreduce_sum_f32": (_I, [_V, _V, _SZ, _V]),
    "linna_allgather_f32": (_I, [_V, _V, _V, _SZ, _V]),
    "linna_broadcast_f32": (_I, [_V, _V, _SZ, _I, _V]),
    "linna_gemm_f32": (_I, [_V, C.POINTER(Gemm), _V]),
    "linna_gemm_dot_slots": (_I, [_I, _I]),
    "linna_linear_fwd": (_I, [_V, _V, _I, _V, _I, _V, _V, _I, _I, _I, _I, _I, _F, _V, _I, _V]),
    "linna_resblock_fwd": (_I, [_V, _V, _I, _V, _V, _V, _V, _V, _V, _I, _V, _I, _I, _I, _I, _I, _V]),
    "linna_linear_bwd": (_I, [_V, _V, _I, _V, _I, _V, _I, _V, _I, _V, _I, _V, _I, _V, _I, _I, _I, _F, _V]),
    "linna_net_create": (_I, [_V, C.POINTER(Layer), _I, _I, _PV]),
    "linna_net_destroy": (_I, [_V]),
    "linna_net_prepare": (_I, [_V, _I, _I]),
    "linna_net_fwd_ws_bytes": (_SZ, [_V, _I]),
    "linna_net_bwd_ws_bytes": (_SZ, [_V, _I]),
    "linna_net_forward": (_I, [_V, _V, _I, _I, _V, _V, _I, C.POINTER(ColMap), _V]),
    "linna_net_backward": (_I, [_V, _V, _I, _I, _V, _V, _V, _I, _V, _I, _I, _V]),
    "linna_net_stream_state": (_I, [_V, _V, _V, _V]),
    "linna_prior_map_fwd": (_I, [_V, _V, _I, _I, _I, _V, _V, _V, _V, _V, _V, _V, _I, _V, _I, _V]),
    "linna_prior_map_bwd": (_I, [_V, _V, _I, _I, _I, _V, _V, _V, _V, _V, _V, _I, _V, _I, _V]),
    "linna_gauss_loglike_diag": (_I, [_V, _V, _I, _I, _I, _V, _V, _I, _I, _F, _V, _V]),
    "linna_gauss_loglike_dense": (_I, [_V, _V, _I, _I, _I, _V, _I, _V, _I, _I, _F, _V, _V, _V]),
    "linna_logprob_create": (_I, [_V, _V, C.POINTER(LogprobDesc), _PV]),
    "linna_logprob_destroy": (_I, [_V]),
    "linna_weights_changed": (_I, [_V]),
    "linna_engine_rows": (_I, [_I]),
    "linna_dense_tri": (_I, [_I]),
    "linna_slice_fusion": (_I, [_I]),
    "linna_net_train_launches": (_I, [_V, _I]),
    "linna_program_describe": (_I, [_V, _I, _I, _I, _I, _V, C.c_size_t]),
    "linna_logprob_ws_bytes": (_SZ, [_V, _I, _I]),
    "linna_logprob_eval": (_I, [_V, _V, _I, _I, _V, _V, _V, _I, _V]),
    "linna_logprob_grad": (_I, [_V, _V, _I, _I, _V, _V, _V, _I, _V]),
    "linna_logprob_grad_leapfrog": (_I, [_V, _V, _I, _I, _V, _V, _V, _I, _V, _I, _V, _F, _F, _V]),
    "linna_loss_scratch_bytes": (_SZ, [_I, _I]),
    "linna_chi2_md": (_I, [_V, C.POINTER(LossDesc), _V, _I, _I, _V, _V, _V]),
    "linna_chi2_ratio_loss_fwd_bwd": (_I, [_V, C.POINTER(LossDesc), _V, _I, _V, _I, _V, _V, _I, _V, _V, _V, _V, _I, _F, _V]),
    "linna_net_prepare_loss": (_I, [_V, C.POINTER(LossDesc)]),
    "linna_loss_targets": (_I, [_V, C.POINTER(LossDesc), _V, _I, _I, _V, _I, _V]),
    "linna_net_forward_loss": (_I, [_V, C.POINTER(LossDesc), _V, _I, _V, _I, _V, _V, _V, _V, _I, _V, _V, _I, _V, _I, _V, _F, _V, _V, _V, _I,
                                    _V, _V, _F, _F, _V]),
    "linna_net_train_step": (_I, [_V, C.POINTER(LossDesc), _V, _I, _V, _I, _V, _V, _V, _V, _I, _V, _V, _I, _V, _I, _V, _F, _V, _V, _V, _I,
                                  _V, _V, _V, _F, _F, _V]),
    "linna_net_train_step_update": (_I, [_V, C.POINTER(LossDesc), _V, _I, _V, _I, _V, _V, _V, _V, _I, _V, _V, _I, _V, _I, _V, _F, _V, _V, _V, _I,
                                         _V, _V, _V, _V, _SZ, _V, _V, _F, _F, _F, _V]),
    "linna_val_rows": (_I, [_V, C.POINTER(LossDesc), _V, _I, _V, _I, _V, _I, _V, _V, _V, _V]),
    "linna_val_metrics": (_I, [_V, _V, _V, _I, _V, _V, _V]),
    "linna_gather_xform": (_I, [_V, _V, _I, _V, _I, _I, _V, _V, _V, _V, _I, _V]),
    "linna_adamw_step": (_I, [_V, _V, _V, _V, _V, _SZ, _V, _V, _F, _F, _F, _I, _V]),
    "linna_net_adamw_step": (_I, [_V, _I, _V, _V, _V, _V, _SZ, _V, _V, _F, _F, _F, _I, _V]),
    "linna_stretch_propose": (_I, [_V, _V, _I, _I, _V, _I, _V, _I, _V, _I, _U64, _V, _I, _F, _V, _I, _V, _V]),
    "linna_stretch_accept": (_I, [_V, _V, _I, _I, _V, _V, _I, _V, _I, _V, _V, _U64, _V, _I, _V, _V]),
    "linna_logprob_eval_if": (_I, [_V, _V, _I, _I, _V, _V, _V, _I, _V, _V]),
    "linna_logprob_eval_slice_points": (_I, [_V, _V, _I, _I, _V, _I, _V, _I, _V, _I, _V, _V, _V]),
    "linna_stretch_half_step": (_I, [_V, _V, _I, _I, _V, _V, _I, _V, _I, _V, _I, _U64, _V, _I, _I, _F, _V, _V]),
    "linna_stretch_run": (_I, [_V, _V, _I, _I, _V, _I, _V, _I, _I, _U64, _V, _I, _F, _V, _V, _V, _V]),
    "linna_chain_append_t": (_I, [_V, _V, _I, _I, _I, _I, _I, _V, _I, _I64, _V]),
    "linna_acorr_update": (_I, [_V, _V, _I, _I, _I, _I64, _I64, _I64, _I64, _I, _I, _V, _V, _I, _V]),
    "linna_acorr_scratch_bytes": (_SZ, [_I, _I, _I]),
    "linna_acorr_tau": (_I, [_V, _V, _I, _I, _I, _I, _I64, _I64, _I, _V, _V, _D, _V, _V, _V]),
    "linna_chain_meanstd": (_I, [_V, _V, _I, _I, _I, _I64, _I64, _I64, _V, _V]),
    "linna_hmc_init": (_I, [_V, _I, _I, _V, _U64, _V, _V, _V, _I, _V, _I, _V, _V]),
    "linna_hmc_start": (_I, [_V, _I, _I, _V, _U64, _V, _V, _V, _I, _V, _I, _F, _F, _V, _I, _V, _I, _V, _I, _V, _V]),
    "linna_hmc_kick_drift": (_I, [_V, _I, _I, _V, _F, _F, _V, _I, _V, _I, _V, _I, _V]),
    "linna_hmc_accept": (_I, [_V, _I, _I, _V, _U64, _V, _V, _V, _I, _V, _I, _V, _V, _I, _V, _V, _I, _V, _V, _V, _V]),
    "linna_step_increment": (_I, [_V, _V, _V]),
    "linna_slice_init": (_I, [_V, _V, _V, _I, _V, _I, _V, _I, _I, _V, _U64, _V, _I, _V, _I, _V, _V, _V, _V, _I, _V]),
    "linna_slice_points": (_I, [_V, _V, _I, _I, _V, _I, _V, _I, _V, _V, _I, _I, _V]),
    "linna_slice_expand": (_I, [_V, _V, _V, _V, _V, _V, _V, _I, _V, _I, _V]),
    "linna_slice_draw": (_I, [_V, _V, _V, _V, _V, _V, _I, _U64, _V, _I, _I, _I, _V]),
    "linna_slice_shrink": (_I, [_V, _V, _V, _V, _V, _V, _V, _V, _V, _I, _V, _I, _I, _V]),
    "linna_slice_commit": (_I, [_V, _V, _I, _I, _V, _V, _I, _V, _I, _V, _V, _V]),
    "linna_slice_half_step": (_I, [_V, _V, _I, _I, _V, _V, _I, _V, _I, _V, _I, _V, _U64, _V, _I, _V, _I, _V, _I, _V, _I, _V, _V, _V, _V, _V, _V, _V, _I, _I, _V, _I, _V]),
}
EXPORTED = tuple(_SIGNATURES)

_lib = None


class LinnaHipError(RuntimeError):
    pass


def load():
    """dlopen the library once; raises LinnaHipError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LinnaHipError("liblinna_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.linna_abi_version() != ABI_VERSION:
        raise LinnaHipError("ABI version mismatch: library %d, binding %d" % (lib.linna_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


ERR_UNSUPPORTED = -3     # LINNA_ERR_UNSUPPORTED (include/linna_hip.h)
ERR_INTERNAL = -4        # LINNA_ERR_INTERNAL: a C++ exception caught at the C boundary


def check(rc):
    if rc != 0:
        raise LinnaHipError("liblinna_hip error %d: %s" % (rc, load().linna_last_error().decode()))


def call(name, *args):
    check(getattr(load(), name)(*args))


_ctx = {}


def ctx(device_index=None):
    """Per-device context handle (creating it verifies that the device is gfx950)."""
    if not torch.cuda.is_available():
        raise LinnaHipError("no HIP device visible: the LINNA hot path has no CPU fallback")
    if device_index is None:
        device_index = torch.cuda.current_device()
    if device_index not in _ctx:
        h = C.c_void_p()
        call("linna_ctx_create", int(device_index), C.byref(h))
        _ctx[device_index] = h
    return _ctx[device_index]


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=torch.float32):
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise LinnaHipError("expected a device tensor")
    if t.dtype != dtype:
        raise LinnaHipError("expected dtype %s, got %s" % (dtype, t.dtype))
    if not t.is_contiguous():
        raise LinnaHipError("expected a contiguous tensor")
    return C.c_void_p(t.data_ptr())


def iptr(t):
    return ptr(t, torch.int32)


def ld4(w):
    return (int(w) + 3) & ~3


class _Stage(object):
    __slots__ = ("name", "t0")

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if stage_profile.current is not None:
            import time
            self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        d = stage_profile.current
        if d is not None:
            import time
            d[self.name] = d.get(self.name, 0.0) + time.perf_counter() - self.t0
        return False


def stage(name):
    """``with stage("theory"):`` -- wall seconds of a stage of a whole run (``ml_sampler_core``: main.py:139-334), added to the
    dict a caller installed with ``stage_profile``; without one the block costs two attribute reads.  Stages nest (an inner
    stage's seconds are also inside its parent's): names carry their parent as a prefix."""
    return _Stage(name)


class stage_profile(object):
    """``with stage_profile(d):`` -- every ``stage(...)`` block inside adds its wall seconds to ``d`` (bench.py's ``e2e``)."""
    current = None

    def __init__(self, d):
        self.d = d

    def __enter__(self):
        self.prev, stage_profile.current = stage_profile.current, self.d
        return self.d

    def __exit__(self, *exc):
        stage_profile.current = self.prev
        return False


class quiet_gc(object):
    """Around a loop that keeps the GPU fed from the host: every object alive now is put aside (``gc.freeze``: constant time;
    no collection first -- that alone held a 1.5 s run for 0.15 s) so that the cyclic collector's passes inside the loop look
    at the loop's own garbage only.  A full pass over an interpreter with torch and numpy loaded holds the thread for
    50-70 ms; the launches queued ahead of it last a few ms -- three such passes were 13 % of a 20 000-iteration run at 128
    walkers (bench.py `host_gc_s`).  ``gc.unfreeze`` at the end hands everything back to the collector."""

    def __enter__(self):
        import gc
        self.on = gc.isenabled() and os.environ.get("LINNA_QUIET_GC", "1") != "0"      # (LINNA_QUIET_GC=0: A/B)
        if self.on:
            gc.freeze()
        return self

    def __exit__(self, *exc):
        import gc
        if self.on:
            gc.unfreeze()
        return False


def slice_fusion(mask=-1):
    """Which launches of linna_slice_half_step are folded into their neighbours (a mask, include/linna_hip.h); -1 queries.
    Returns the previous mask (tests and measurements: the chain is the same under every mask)."""
    prev = load().linna_slice_fusion(int(mask))
    if prev < 0:
        check(prev)
    return prev


def engine_rows(rows=0):
    """Force one engine of the whole-network kernel (4, 8 or 16 rows per workgroup) for every later launch of this
    process; 0 = chosen per launch from the batch size.  Returns the previous setting (tests and measurements)."""
    prev = load().linna_engine_rows(int(rows))
    if prev < 0:
        check(prev)
    return prev
