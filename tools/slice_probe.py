"""Timing/profile target: zeus-style ensemble slice sampler iterations on the bench problem."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from linna_amd import sampler
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lp, model, consts = bench.build_problem(torch.device("cuda", 0))
ens = sampler.SliceEnsembleSampler(nw, 33, lp, seed=1)
ens.set_state(0.05 * np.random.RandomState(7).standard_normal((nw, 33)))
ens.run(40, store=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 40
ens.run(n, store=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("slice sampler: %d walkers, %.1f us/iteration, %.0f it/s, mu %.3f, evals/iter/walker %.2f" % (nw, dt / n * 1e6, n / dt, ens.mu, getattr(ens, "neval", 0) / max(1, ens.iteration) / nw))
