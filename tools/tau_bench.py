"""Cost of one convergence check on the device (csrc/autocorr.hip), kernel by kernel with HIP events: the transposing append
of 100 new chain rows, the running-sum update over K lags, emcee's estimate from the sums -- at the reference's production
ensemble (128 walkers) and at the bench's 4096 walkers, for the lag capacities a run reaches; and the achieved float64
FMA rate of the update against the vector pipe's peak (256 CUs x 64 FMA/clk x 2.4 GHz = 39.3 TFMA/s = 78.6 TFLOP/s).
usage: tau_bench.py [nwalkers lags]..."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from linna_amd import _lib
from linna_amd.sampler import DeviceChain

PEAK_TFMA = 256 * 64 * 2.4e9 / 1e12


def ev_ms(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


args = [int(a) for a in sys.argv[1:]]
cases = list(zip(args[0::2], args[1::2])) or [(128, 512), (128, 2048), (128, 5120), (4096, 512), (4096, 1024), (4096, 2112), (-4096, 2112)]
nd = 33
for nw, lags in cases:
    nt = lags + 600
    dc = DeviceChain(max_walkers=None) if nw < 0 else DeviceChain()      # negative: every walker in the running sums
    nw = abs(nw)
    dc.LAGS0 = lags
    x = torch.randn(nt, nw, nd, device="cuda").cumsum(0) * 0.01 + torch.randn(nt, nw, nd, device="cuda")
    dc.append(x[:nt - 100])
    dc.integrated_time()
    blk = x[nt - 100:].contiguous()
    n0 = dc.n
    t_app = ev_ms(lambda: _lib.call("linna_chain_append_t", dc.ctx, _lib.ptr(blk), nd, 100, nw, nd, 1, _lib.ptr(dc.ct), dc.nwp, n0, _lib.stream()))
    dc.n = n0 + 100
    S, T = dc._S, dc._T
    t_upd = ev_ms(lambda: dc._update(S, T, n0, n0 + 100, 0, n0 + 100, 0, len(S), False))
    t_rem = ev_ms(lambda: dc._update(S, T, 0, 20, 0, n0 + 100, 0, len(S), True))
    dc._S, dc._T = dc._fresh(len(S))
    dc._lo = dc._hi = 0
    dc._advance(0, dc.n)
    t_tau = ev_ms(lambda: dc._estimate(dc._S, dc._T, 0, dc.n, 5.0, dc.nws))
    fma = dc.nd * dc.nwc * 100.0 * len(S)
    print("nw %5d lags %5d (series %6d, padded %6d): append_t %7.1f us | update(100 rows) %8.1f us = %5.2f TFMA/s = %4.1f %% of the f64 vector peak "
          "| remove(20 rows) %7.1f us | tau from sums %7.1f us | check total %8.1f us" % (
              nw, len(S), dc.nws * nd, dc.nd * dc.nwc, 1e3 * t_app, 1e3 * t_upd, fma / (t_upd * 1e-3) / 1e12, 100 * fma / (t_upd * 1e-3) / 1e12 / PEAK_TFMA,
              1e3 * t_rem, 1e3 * t_tau, 1e3 * (t_app + t_upd + t_tau)), flush=True)
