"""Driver-level sampling rate: sampler.HMCSampler.sample (emcee driver: burn-in, chain blocks to the host, HDF5 appends,
convergence checks) against the raw EnsembleSampler.run rate, on the bench problem; where the wall time goes."""
import sys, os, time, tempfile
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
    sampler.DeviceChain.prewarm(nw, 33, torch.device("cuda", 0)).join()      # (ml_sampler_core does this while the emulator trains)
    out = tempfile.mkdtemp(dir=os.environ.get("LINNA_PROBE_DIR"))
    drv = sampler.HMCSampler(lp, None, None, 33, nw, x0=x0, transform=util.Transform(priors))
    nsamp = int(os.environ.get("LINNA_PROBE_NSAMP", "3000" if nw > 1000 else "20000"))
    marks = {}
    orig_flush = sampler.ChainStore.flush
    def flush(self, final=True):
        t = time.perf_counter(); r = orig_flush(self, final); marks[final] = marks.get(final, 0.0) + time.perf_counter() - t; return r
    sampler.ChainStore.flush = flush
    orig_it = sampler.DeviceChain.integrated_time
    def it(self, *a, **k):
        t = time.perf_counter(); r = orig_it(self, *a, **k); marks["tau"] = marks.get("tau", 0.0) + time.perf_counter() - t; return r
    sampler.DeviceChain.integrated_time = it
    t0 = time.perf_counter()
    store = drv.sample(None, nsamp, outdir=out, ntimes=1e9, tautol=1e-9, incremental=True)     # never converges: runs nsamp
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    sampler.ChainStore.flush, sampler.DeviceChain.integrated_time = orig_flush, orig_it
    n = sum(len(c) for c in store.chain)
    sz = os.path.getsize(os.path.join(out, "chemcee_256.h5")) / 1e6
    import shutil
    shutil.rmtree(out, ignore_errors=True)
    print("nw %d: raw %.0f it/s (storing the chain on the device: %.0f); driver %d iterations (+100 burn-in) in %.2f s = %.0f it/s; "
          "incremental flushes %.2f s, final flush %.2f s, tau estimates %.2f s; file %.0f MB in %s" % (
              nw, raw, raw_store, n, dt, (n + 100) / dt, marks.get(False, 0), marks.get(True, 0), marks.get("tau", 0), sz, out), flush=True)
