"""A/B of diagnostic builds of the library (linna_amd/_build.py build_stamps(extra flags, name)): the headline kernel, the
ChtoModelv2(33,33) serving launch and the stretch half step, each variant in a process of its own.
usage: variant_bench.py <lib.so> [<lib.so> ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os, json
sys.path.insert(0, %r)
import torch, bench
dev = torch.device("cuda", 0)
a = bench.secondary_serving(dev, "MLP", 33, 33, False, 4096, 1500)
b = bench.secondary_serving(dev, "ChtoModelv2", 33, 33, False, 4096, 1200)
c = bench.secondary_serving(dev, "ChtoModelv2", 40, 1000, True, 4096, 500)
lp, model, consts = bench.build_problem(dev)
dt, m = bench.mcmc_rate(lp, 4096, 1, None, 600, 300)
print("%%-28s mlp %%.2f us  v2 %%.2f us  dense1000 %%.2f us  stretch %%.0f it/s" %% (os.path.basename(os.environ.get("LINNA_LIB_PATH", "default")), a["us_per_launch"], b["us_per_launch"], c["us_per_launch"], m["steps_per_s"]))
''' % ROOT
for lib in sys.argv[1:]:
    env = dict(os.environ)
    if lib != "default":
        env["LINNA_LIB_PATH"] = os.path.abspath(lib)
    for rep in range(2):
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print((r.stdout.strip().splitlines() or ["(no output) " + r.stderr[-300:]])[-1], flush=True)
