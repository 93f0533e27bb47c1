"""Networks outside the whole-network kernel's reach (a layer wider than 1024: ChtoModelv2(nin, nout > 1024) has
layer8 = nout x nout): the layer-by-layer GEMM path serves them -- evaluation, gradient and a training step against the
oracle.  usage: wide_probe.py [nout ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np, torch
import cases, synth
from test_gpu_serving import build_logprob
from oracle import likelihood

NIN = int(os.environ.get("WIDE_NIN", "12"))
for nout in [int(a) for a in sys.argv[1:]] or [1100, 1500]:
    nin, seed = NIN, 900 + nout
    for dense in (False, True):
        data, cov, priors = synth.gaussian_problem(nin, nout, seed, dense=dense, cond=1e2)
        X_mean, X_std, y_mean, y_std = synth.transform_constants(nin, nout, seed)
        prob = dict(kind="ChtoModelv2", nin=nin, nout=nout, kw={}, weights=synth.weights("ChtoModelv2", nin, nout, seed), priors=priors,
                    data=data, cov=cov, invcov=np.linalg.inv(cov), sigma=np.sqrt(np.diag(cov)), X_mean=X_mean, X_std=X_std, y_mean=y_mean,
                    y_std=y_std, dolog10=None, ypositive=False)
        lp = build_logprob(None, 1.0, prob)[0]
        z = np.random.RandomState(1).standard_normal((300, nin)).astype(np.float32) * 0.5
        got = lp(z, returntorch=False)
        ref = likelihood.log_prob(z, cases.oracle_emulator(prob), priors, data, prob["invcov"], 1.0, dtype=np.float64)
        zd, _ = lp._to_device(z)
        lnp, g = lp.evaluate_with_grad(zd)
        _, gref = likelihood.grad_log_prob(z, cases.oracle_emulator(prob), priors, data, prob["invcov"], 1.0, dtype=np.float64)
        eg = np.abs(g.cpu().numpy() - gref).max() / np.abs(gref).max()
        print("ChtoModelv2(%d,%d) %s: lnP max rel err %.2e (eval), %.2e (grad path); gradient %.2e of max" % (
            nin, nout, "dense" if dense else "diag", np.max(np.abs(got - ref) / np.abs(ref)),
            np.max(np.abs(lnp.cpu().numpy() - ref) / np.abs(ref)), eg), flush=True)
