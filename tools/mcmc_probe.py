"""Timing / trace target: stretch-move ensemble steps (4096 walkers, bench problem)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from linna_amd import sampler
dev = torch.device("cuda", 0)
lp, model, consts = bench.build_problem(dev)
ens = sampler.EnsembleSampler(4096, 33, lp, seed=1)
ens.set_state(0.05 * np.random.RandomState(7).standard_normal((4096, 33)))
ens.run(300, store=False)
torch.cuda.synchronize()
import time
t0 = time.perf_counter(); ens.run(1000, store=False); torch.cuda.synchronize()
print("%.1f us per step" % ((time.perf_counter() - t0) * 1e3))
