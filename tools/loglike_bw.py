#!/usr/bin/env python
"""HBM roofline of the standalone Gaussian log-likelihood reduction (linna_gauss_loglike_diag: one wavefront per
walker row, float4 reads, 64-lane shuffle reduction; util.py:953-955 with a diagonal covariance).  Algorithmic
bytes per evaluation = 4*(nout + nin) + 4; sizes well past the 256 MiB Infinity Cache so that the reads are HBM."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linna_amd import _lib
PEAK = 8000.0   # GB/s, MI355X_MICROARCH.md
for B, nout, nin in [(1 << 20, 1000, 40), (1 << 21, 457, 26), (1 << 22, 33, 33), (1 << 22, 2, 2), (1000003, 125, 7)]:
    ld = _lib.ld4(nout)
    D = torch.randn(B, ld, device="cuda"); w = torch.rand(nout, device="cuda") + 0.5
    Z = torch.randn(B, _lib.ld4(nin), device="cuda"); out = torch.empty(B, device="cuda")
    ctx, st = _lib.ctx(), _lib.stream()
    run = lambda: _lib.call("linna_gauss_loglike_diag", ctx, _lib.ptr(D), ld, B, nout, _lib.ptr(w), _lib.ptr(Z), Z.stride(0), nin,
                            C.c_float(1.0), _lib.ptr(out), st)
    for _ in range(3): run()
    e0, e1 = C.c_void_p(), C.c_void_p()
    _lib.call("linna_event_create", C.byref(e0)); _lib.call("linna_event_create", C.byref(e1))
    _lib.call("linna_event_record", e0, st)
    n = 10
    for _ in range(n): run()
    _lib.call("linna_event_record", e1, st)
    ms = C.c_float(); _lib.call("linna_event_elapsed_ms", e0, e1, C.byref(ms))
    torch.cuda.synchronize(); sel = torch.tensor([0, 1, 2, B // 2, B - 2, B - 1], device="cuda")
    ref = (-0.5 * (D[sel][:, :nout].double() ** 2 * w.double()).sum(1) - 0.5 * (Z[sel][:, :nin].double() ** 2).sum(1)).cpu().numpy()
    assert np.allclose(out[sel].cpu().numpy(), ref, rtol=1e-4)
    byts = B * (4.0 * (nout + nin) + 4)
    gbs = byts / (ms.value / n * 1e-3) / 1e9
    print("B=%d nout=%d: %.1f us, %.0f GB/s algorithmic = %.1f %% of %.0f GB/s (%.2f GB per launch)" % (
        B, nout, ms.value / n * 1e3, gbs, 100 * gbs / PEAK, PEAK, byts / 1e9), flush=True)
