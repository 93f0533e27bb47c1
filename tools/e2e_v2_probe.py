"""Probe: ml_sampler_core through ChtoModelv2(26, 457) on the closed-form linear problem (tests/golden/linear457.py)."""
import os, sys, time, shutil, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import linear457
from linna_amd.main import ml_sampler_core
from linna_amd import nn

nw = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nep = int(sys.argv[2]) if len(sys.argv) > 2 else 400
ntrain = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
prob = linear457.problem()
sig = np.sqrt(np.diag(prob["post_cov"]))
print("posterior sigma / prior half width: min %.3f max %.3f" % ((sig / (0.5 * (prob["hi"] - prob["lo"]))).min(), (sig / (0.5 * (prob["hi"] - prob["lo"]))).max()))
out = tempfile.mkdtemp(prefix="linna_v2_") + "/"
np.random.seed(0); torch.manual_seed(linear457.SEED)
t0 = time.time()
params = {"trainingoption": 1, "num_epochs": nep, "batch_size": 500}
chain, logp = ml_sampler_core([ntrain] * 4, [500] * 4, [2, 2, 5, 4], [5, 5, 10, 15], [0.03, 0.03, 0.02, 0.01], [0.2] * 4, [0.15] * 4, out,
                              linear457.Theory(prob["A"], prob["c"]), prob["priors"], prob["data"], prob["cov"], prob["init"], None, nw, "cuda",
                              [0, 1], False, [4.0, 2.0, 1.0, 1.0], None, False, 1, None, nn.ChtoModelv2, params, "emcee")
print("wall %.1f s, chain %s" % (time.time() - t0, chain.shape))
print("bias %.4f  std %.4f  corr %.4f" % linear457.summary(chain, prob))
lp = np.asarray(logp).reshape(-1)[-len(chain):]
from linna_amd import util
z = util.invTransform(prob["priors"])(chain[:20000])
d = chain[:20000] @ prob["A"].T + prob["c"] - prob["data"]
exact = -0.5 * np.einsum("bi,ij,bj->b", d, prob["icov"], d) - 0.5 * np.sum(np.asarray(z) ** 2, axis=1)
err = lp[:20000] - exact
print("stored logp - exact: median %.3f  p1 %.3f p99 %.3f  max|.| %.3f" % (np.median(err), np.percentile(err, 1), np.percentile(err, 99), np.abs(err).max()))
for k in range(4):
    print(k, os.path.getsize(os.path.join(out, "iter_%d" % k, "chemcee_256.h5")) >> 20, "MB")
shutil.rmtree(out, ignore_errors=True)
