"""A/B of diagnostic builds on the workloads of the 4x4x1 (L1-fill-bound) engines: the training step at (26,457) and (33,33),
the stretch iterations at 4096 and 128 walkers, an evaluation of 1024 / 2048 walkers.  Each variant in a process of its own.
usage: variant_bench_small.py <lib.so|default> ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os, time
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
import numpy as np, torch, bench
dev = torch.device("cuda", 0)
tr = bench.training_rate(dev, 1, 0, "nccl")
lp, model, consts = bench.build_problem(dev)
_, m4096 = bench.mcmc_rate(lp, 4096, 1, None, 600, 300)
_, m128 = bench.mcmc_rate(lp, 128, 1, None, 3000, 800)
out = {}
for B in (1024, 2048):
    z = torch.as_tensor(np.random.RandomState(5).standard_normal((B, 33)).astype(np.float32), device=dev)
    o = torch.empty(B, dtype=torch.float32, device=dev)
    out[B] = bench._events_us(lambda: lp.evaluate(z, out=o), 600)
print("%%-30s train %%.1f us  stretch4096 %%.0f it/s  stretch128 %%.0f it/s  eval1024 %%.2f us  eval2048 %%.2f us" %% (
    os.path.basename(os.environ.get("LINNA_LIB_PATH", "default")), 1e3 * tr["ms_per_step"], m4096["steps_per_s"], m128["steps_per_s"], out[1024], out[2048]))
''' % (ROOT, ROOT)
for lib in sys.argv[1:]:
    env = dict(os.environ)
    if lib != "default":
        env["LINNA_LIB_PATH"] = os.path.abspath(lib)
    for rep in range(2):
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print((r.stdout.strip().splitlines() or ["(no output) " + r.stderr[-400:]])[-1], flush=True)
