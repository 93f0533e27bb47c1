"""End-to-end timing of ml_sampler on the README problem (33-D Gaussian, theory = identity):
where does the wall time go between point generation, training and sampling?"""
import sys, os, time, tempfile, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from linna_amd.main import ml_sampler
np.random.seed(0)
ndim = 33
means = np.random.uniform(size=ndim)
cov = np.diag(0.1 * np.random.uniform(size=ndim))
init = np.random.uniform(size=ndim)
priors = [{"param": "p%d" % i, "dist": "flat", "arg1": -5.0, "arg2": 5.0} for i in range(ndim)]
def theory(x, outdir):
    return x[1]
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nepoch = int(sys.argv[2]) if len(sys.argv) > 2 else 101
out = tempfile.mkdtemp()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
chain, logp = ml_sampler(out, theory, priors, means, cov, init, None, nw, gpunode=None, nepoch=nepoch, method="emcee")
pr.disable()
dt = time.perf_counter() - t0
th = np.asarray(chain)
print("ml_sampler: %d walkers, nepoch %d: %.1f s; chain %s; max |mean-means|/sigma %.3f" % (
    nw, nepoch, dt, th.shape, np.max(np.abs(th.mean(0) - means) / np.sqrt(np.diag(cov)))))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
