"""Driver-level sampling rate: sampler.HMCSampler.sample (emcee driver: burn-in, chain blocks to the host, HDF5 appends,
convergence checks at every 100 iterations) against the raw EnsembleSampler.run rate, on the bench problem; where the wall
time goes (the driver's own profile: host seconds and device seconds per phase).
usage: driver_probe.py [nwalkers ...]; LINNA_PROBE_DIR = directory of the chain file (default: the temporary directory),
LINNA_PROBE_NSAMP = iterations."""
import sys, os, time, tempfile, json, shutil, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from linna_amd import sampler, util
lp, model, consts = bench.build_problem(torch.device("cuda", 0))
priors = [{"param": "p%d" % i, "dist": "flat", "arg1": -5.0, "arg2": 5.0} for i in range(33)]
sizes = [int(a) for a in sys.argv[1:]] or [4096, 128]
for nw in sizes:
    ens = sampler.EnsembleSampler(nw, 33, lp, seed=1)
    x0 = 0.05 * np.random.RandomState(7).standard_normal((nw, 33))
    ens.set_state(x0); ens.run(300, store=False); torch.cuda.synchronize()
    t0 = time.perf_counter(); ens.run(1000, store=False); torch.cuda.synchronize()
    raw = 1000 / (time.perf_counter() - t0)
    t0 = time.perf_counter(); ens.run(1000, store=True); torch.cuda.synchronize()
    raw_store = 1000 / (time.perf_counter() - t0)
    ens.block_run = False
    t0 = time.perf_counter(); ens.run(1000, store=True); torch.cuda.synchronize()
    raw_loop = 1000 / (time.perf_counter() - t0)
    out = tempfile.mkdtemp(dir=os.environ.get("LINNA_PROBE_DIR"))
    drv = sampler.HMCSampler(lp, None, None, 33, nw, x0=x0, transform=util.Transform(priors))
    nsamp = int(os.environ.get("LINNA_PROBE_NSAMP", "3000" if nw > 1000 else "20000"))
    prof = {}
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        store = drv.sample(None, nsamp, outdir=out, ntimes=1e9, tautol=1e-9, incremental=True, profile=prof)     # never converges: runs nsamp
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n = sum(len(c) for c in store.chain)
    sz = os.path.getsize(os.path.join(out, "chemcee_256.h5")) / 1e6
    shutil.rmtree(out, ignore_errors=True)
    print("nw %d: raw %.0f it/s (one C call per block, chain rows from the kernels: %.0f; per-iteration loop + copies: %.0f); driver %d "
          "iterations (+100 burn-in) in %.2f s = %.0f it/s; file %.0f MB in %s" % (nw, raw, raw_store, raw_loop, n, dt, (n + 100) / dt, sz, out), flush=True)
    print("   profile: " + json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in sorted(prof.items())}), flush=True)
