import sys, os, time, tempfile, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from linna_amd import sampler, util
lp, model, consts = bench.build_problem(torch.device("cuda", 0))
priors = [{"param": "p%d" % i, "dist": "flat", "arg1": -5.0, "arg2": 5.0} for i in range(33)]
nw = 4096
x0 = 0.05 * np.random.RandomState(7).standard_normal((nw, 33))
drv = sampler.HMCSampler(lp, None, None, 33, nw, x0=x0, transform=util.Transform(priors))
pr = cProfile.Profile(); pr.enable()
store = drv.sample(None, 1500, outdir=tempfile.mkdtemp(), ntimes=1e9, tautol=1e-9, incremental=True)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
