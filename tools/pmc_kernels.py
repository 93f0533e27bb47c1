"""Workload for a rocprofv3 --pmc pass over the variants of the whole-network kernel: evaluation on the 16-, 8- and
4-row engines, the fused stretch half step, the fused gradient, dense covariance (40,1000), training forward and
dX chain.  A few hundred launches each after a warm-up (summarise with tools/pmc_mfma.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.argv = sys.argv[:1]
import numpy as np, torch, bench_paths
from bench_paths import problem
from linna_amd import sampler
N = 300
def loop(fn, n=N):
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
p = problem("MLP", 33, 33, False, width=512, depth=4)
lp = p["lp"]
for B in (4096, 2048, 512):
    z = torch.randn(B, 33, device="cuda"); o = torch.empty(B, device="cuda"); g = torch.empty(B, 33, device="cuda")
    loop(lambda: lp.evaluate(z, out=o), 600)
    loop(lambda: lp.evaluate_with_grad(z, out=o, grad=g), 300)
ens = sampler.EnsembleSampler(4096, 33, lp, seed=1)
ens.set_state(0.05 * np.random.RandomState(7).standard_normal((4096, 33)))
ens.run(300, store=False); torch.cuda.synchronize()
pd = problem("MLP", 40, 1000, True, width=512, depth=4)
z = torch.randn(4096, 40, device="cuda"); o = torch.empty(4096, device="cuda")
loop(lambda: pd["lp"].evaluate(z, out=o), 300)
pv = problem("ChtoModelv2", 33, 33, False)
z = torch.randn(4096, 33, device="cuda"); o = torch.empty(4096, device="cuda")
loop(lambda: pv["lp"].evaluate(z, out=o), 300)
m = pv["model"]; x = torch.randn(500, 33, device="cuda"); d = torch.randn(500, 33, device="cuda")
def fb():
    m.forward(x); m.backward(d, param_grads=True)
loop(fb, 200)
