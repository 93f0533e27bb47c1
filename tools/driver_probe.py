"""Driver-level sampling rate: sampler.HMCSampler.sample (emcee driver: chain to host, part files, convergence
checks) against the raw EnsembleSampler.run rate, on the bench problem."""
import sys, os, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from linna_amd import sampler, util
lp, model, consts = bench.build_problem(torch.device("cuda", 0))
priors = [{"param": "p%d" % i, "dist": "flat", "arg1": -5.0, "arg2": 5.0} for i in range(33)]
for nw in (4096, 128):
    ens = sampler.EnsembleSampler(nw, 33, lp, seed=1)
    x0 = 0.05 * np.random.RandomState(7).standard_normal((nw, 33))
    ens.set_state(x0); ens.run(300, store=False); torch.cuda.synchronize()
    t0 = time.perf_counter(); ens.run(1000, store=False); torch.cuda.synchronize()
    raw = 1000 / (time.perf_counter() - t0)
    out = tempfile.mkdtemp()
    drv = sampler.HMCSampler(lp, None, None, 33, nw, x0=x0, transform=util.Transform(priors))
    nsamp = 3000 if nw > 1000 else 20000
    t0 = time.perf_counter()
    store = drv.sample(None, nsamp, outdir=out, ntimes=1e9, tautol=1e-9, incremental=True)     # never converges: runs nsamp
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n = sum(len(c) for c in store.chain)
    print("nw %d: raw %.0f it/s; driver %d iterations (+100 burn-in) in %.2f s = %.0f it/s" % (nw, raw, n, dt, (n + 100) / dt), flush=True)
