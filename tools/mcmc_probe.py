"""Profile target: 40 ensemble iterations of the bench problem (4096 walkers)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from linna_amd import sampler
lp, model, consts = bench.build_problem(torch.device("cuda", 0))
ens = sampler.EnsembleSampler(4096, 33, lp, seed=1)
ens.set_state(0.05 * np.random.RandomState(7).standard_normal((4096, 33)))
ens.run(50, store=False)
torch.cuda.synchronize()
