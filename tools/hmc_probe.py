"""Timing target: one gradient launch against one gradient launch with the leapfrog's kick and drift in its finish
(linna_logprob_grad_leapfrog), and whole HMC transitions fused / unfused.  usage: hmc_probe.py [MLP|ChtoModelv2] [chains]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from linna_amd import _lib, sampler
kind = sys.argv[1] if len(sys.argv) > 1 else "MLP"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda", 0)
lp, model = bench._problem33(dev, kind)
x0 = 0.05 * np.random.RandomState(1).standard_normal((B, 33)).astype(np.float32)
for fused in (False, True):
    h = sampler.BatchedHMC(lp, x0, fused=fused)
    us = bench._events_us(lambda: h.step(5, 2e-2), 60)
    print("%s %d chains  HMC transition of 5 leapfrog steps, %s: %.1f us" % (kind, B, "kick + drift in the gradient launch" if fused else "separate launches", us))
h = sampler.BatchedHMC(lp, x0)
hd, ws = lp._ensure()["handle"], _lib.ptr(lp._workspace(B, True))
st = _lib.stream()
plain = lambda: lp.evaluate_with_grad(h.q, out=h.lnp_new, grad=h.g_new)
leap = lambda: _lib.call("linna_logprob_grad_leapfrog", hd, _lib.ptr(h.q), h.ld, B, ws, _lib.ptr(h.lnp_new), _lib.ptr(h.g_new), h.ld,
                         _lib.ptr(h.p), h.ld, _lib.ptr(h.mass), 1e-9, 1e-9, st)
print("gradient launch %.1f us; with kick + drift in its finish %.1f us" % (bench._events_us(plain, 300), bench._events_us(leap, 300)))
