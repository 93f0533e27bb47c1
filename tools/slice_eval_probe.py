"""Timing target: the slice sampler's trial-point evaluation (linna_logprob_eval_slice_points, MOVE == 2 prologue of the
whole-network kernel) against the plain evaluation of the same number of rows.  usage: slice_eval_probe.py [ns] [nrep] [scale]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from linna_amd import _lib
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 16
scales = [float(a) for a in sys.argv[3:]] or [0.0, 0.01, 1.0, 30.0]
dev = torch.device("cuda", 0)
lp, model, consts = bench.build_problem(dev)
nw, ndim, ld = 2 * ns, 33, 36
g = torch.Generator(device="cpu").manual_seed(3)
coords = torch.zeros(nw, ld, device=dev); coords[:, :ndim] = 0.05 * torch.randn(nw, ndim, generator=g).to(dev)
DIR = torch.zeros(ns, ld, device=dev); DIR[:, :ndim] = 0.05 * torch.randn(ns, ndim, generator=g).to(dev)
S = torch.randperm(nw, generator=g)[:ns].to(dev).int()
out = torch.zeros(nrep * ns, device=dev)
P, I, st = _lib.ptr, _lib.iptr, _lib.stream()
h = lp._ensure()["handle"]

def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for sc in scales:
    w = (sc * torch.randn(nrep * ns, generator=g)).to(dev)
    def pts():
        _lib.check(_lib.load().linna_logprob_eval_slice_points(h, P(coords), ld, ndim, I(S), ns, P(DIR), ld, P(w), nrep, P(out), None, st))
    t_pts = timed(pts)
    k = torch.arange(nrep * ns, device=dev) % ns
    Q = torch.zeros(nrep * ns, ld, device=dev)
    Q[:, :ndim] = coords[S.long()[k], :ndim] + w[:, None] * DIR[k, :ndim]
    t_plain = timed(lambda: lp.evaluate(Q))
    ref = lp.evaluate(Q)
    same = bool(torch.equal(ref, out) or torch.allclose(ref, out, rtol=0, atol=0, equal_nan=True))
    print("ns %d nrep %d (%d rows) scale %-6g trial points %8.1f us   plain rows %8.1f us   identical %s   finite %.2f" % (
        ns, nrep, nrep * ns, sc, t_pts, t_plain, same, float(torch.isfinite(ref).float().mean())), flush=True)
