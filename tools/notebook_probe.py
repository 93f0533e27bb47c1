import os, sys, time, io, contextlib, tempfile, shutil
ROOT="/root/repo"; sys.path.insert(0, ROOT)
import numpy as np, torch
from linna_amd import sampler, util, nn, _lib
fix = os.path.join(ROOT, "tests", "golden", "2dgaussian_Fulltconn", "iter_0/")
model, yinv = util.retrieve_model(fix, 2, 2, nn.ChtoModelv2)
priors = [{"param": "test_%d" % i, "dist": "flat", "arg1": -2.0, "arg2": 2.0} for i in range(2)]
data, cov = np.array([0.1, 1.0]), np.diag([0.5, 0.2])
lp = util.Log_prob(data, np.linalg.inv(cov), model, yinv, util.Transform(priors), 1.0, util.gaussianlogliklihood, nograd=True)
for nw in (4, 8, 16, 64):
    x0 = 1e-3 * np.random.RandomState(0).standard_normal((nw, 2))
    ens = sampler.SliceEnsembleSampler(nw, 2, lp, seed=1)
    ens.set_state(x0)
    ens.run(300, store=False); torch.cuda.synchronize()
    e0=ens.neval; t0=time.perf_counter(); n=1000
    ens.run(n, store=False); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print(nw, "raw slice it/s %.0f us/it %.1f" % (n/dt, 1e6*dt/n), "tune", ens.tune, "mu", ens.mu, "fast_ok", ens._fast_ok, "fast_steps", getattr(ens,'_fast_steps',0), "overflow", ens.noverflow, "evals/w/it", (ens.neval-e0)/n/nw, "sched", ens.m_sched, ens.nt_sched, ens.round_usage(), flush=True)
    em = sampler.EnsembleSampler(nw, 2, lp, seed=1); em.set_state(x0); em.run(300, store=False); torch.cuda.synchronize()
    t0=time.perf_counter(); em.run(n, store=False); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print(nw, "raw stretch it/s %.0f" % (n/dt), flush=True)
