#!/usr/bin/env python
"""Latency of the small GEMMs of one training step (batch 500): no K split (flag 0x80) vs the
intra-workgroup K split picked automatically for grids that leave CUs idle.  Shapes (M, N, K, alay, blay)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linna_amd import _lib


def time_gemm(M, N, K, alay, blay, flags, iters=50):
    pad = lambda n: (n + 3) & ~3                      # rows padded to 16 bytes as the library's own buffers are
    A = torch.randn((M, pad(K)) if alay == 0 else (K, pad(M)), device="cuda")
    W = torch.randn((N, pad(K)) if blay == 0 else (K, pad(N)), device="cuda")
    out = torch.empty((M, _lib.ld4(N)), device="cuda")
    g = _lib.Gemm()
    g.npairs, g.alpha0, g.M, g.N = 1, 1.0, M, N
    g.p[0].A, g.p[0].lda, g.p[0].alay = A.data_ptr(), A.stride(0), alay
    g.p[0].B, g.p[0].ldb, g.p[0].blay, g.p[0].K = W.data_ptr(), W.stride(0), blay, K
    g.C, g.ldc, g.relu, g.flags = out.data_ptr(), out.stride(0), 0, flags
    ctx, st = _lib.ctx(), _lib.stream()
    for _ in range(5):
        _lib.call("linna_gemm_f32", ctx, C.byref(g), st)
    e0, e1 = C.c_void_p(), C.c_void_p()
    _lib.call("linna_event_create", C.byref(e0)); _lib.call("linna_event_create", C.byref(e1))
    _lib.call("linna_event_record", e0, st)
    for _ in range(iters):
        _lib.call("linna_gemm_f32", ctx, C.byref(g), st)
    _lib.call("linna_event_record", e1, st)
    ms = C.c_float()
    _lib.call("linna_event_elapsed_ms", e0, e1, C.byref(ms))
    return ms.value / iters * 1e3


shapes = [(500, 1000, 33, 0, 0), (500, 16, 1000, 0, 0), (500, 500, 1016, 0, 0), (500, 32, 500, 0, 0), (500, 250, 532, 0, 0),
          (500, 500, 128, 0, 0), (500, 33, 500, 0, 0), (500, 33, 33, 0, 0),
          (500, 1000, 500, 0, 1), (500, 1000, 16, 0, 1), (500, 500, 33, 0, 1),
          (500, 1000, 500, 1, 1), (16, 1000, 500, 1, 1), (33, 33, 500, 1, 1), (1000, 33, 500, 1, 1)]
shapes += [(500, 1000, 512, 0, 1), (500, 512, 256, 0, 1), (500, 256, 128, 0, 1), (500, 500, 1016, 0, 0), (500, 250, 544, 0, 0),
           (4096, 16, 1000, 0, 0), (4096, 500, 1016, 0, 0)]
print("%-28s %10s %10s" % ("M, N, K, alay, blay", "no K split", "auto"))
for sh in shapes:
    a = min(time_gemm(*sh, 0x80) for _ in range(3)); b = min(time_gemm(*sh, 0) for _ in range(3))
    print("%-28s %8.1f us %8.1f us" % (sh, a, b), flush=True)
