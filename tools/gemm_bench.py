#!/usr/bin/env python
"""Micro-benchmark of the fused fp32-MFMA GEMM (HIP-event timed, one process, interleaved
rounds): tile configurations and ablations (cdna guide section 5.4 rules 17/24)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linna_amd import _lib  # noqa: E402


def time_gemm(M, N, K, flags, iters=50, relu=1):
    A = torch.randn((M, K), device="cuda")
    W = torch.randn((N, K), device="cuda") / np.sqrt(K)
    b = torch.randn(N, device="cuda")
    out = torch.empty((M, _lib.ld4(N)), device="cuda")
    g = _lib.Gemm()
    g.npairs, g.alpha0, g.M, g.N = 1, 1.0, M, N
    g.p[0].A, g.p[0].lda, g.p[0].alay = A.data_ptr(), K, 0
    g.p[0].B, g.p[0].ldb, g.p[0].blay, g.p[0].K = W.data_ptr(), K, 0, K
    g.C, g.ldc, g.bias0, g.relu, g.flags = out.data_ptr(), out.stride(0), b.data_ptr(), relu, flags
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


if __name__ == "__main__":
    shapes = [(4096, 512, 512), (4096, 512, 33), (4096, 33, 512), (500, 1000, 33), (500, 500, 1000), (500, 33, 500),
              (8192, 512, 512), (16384, 512, 512)]
    ncfg = int(os.environ.get("NCFG", "3"))
    print("%-22s %s" % ("shape", "  ".join("cfg%d us (TF)" % c for c in range(ncfg))))
    for (M, N, K) in shapes:
        row = []
        for c in range(ncfg):
            us = min(time_gemm(M, N, K, c + 1) for _ in range(3))
            row.append("%7.1f (%5.1f)" % (us, 2.0 * M * N * K / us / 1e6))
        print("%-22s %s" % ((M, N, K), "  ".join(row)), flush=True)
    print("ablation on (4096,512,512), us: cfg x [full, noload, nostage, nomfma, noload+nostage]")
    for c in range(ncfg):
        r = [min(time_gemm(4096, 512, 512, (c + 1) | f) for _ in range(3)) for f in (0, 0x10, 0x20, 0x40, 0x30)]
        print("cfg%d " % c + " ".join("%7.1f" % v for v in r), flush=True)
