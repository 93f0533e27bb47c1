#!/usr/bin/env python
"""Headline benchmark: emulator log-likelihood evaluations per second on MI355X.

Workload (BASELINE.json configs[1]): 33-D Gaussian posterior, 4 x 512 ReLU MLP emulator
(33 -> 512 x 4 -> 33, fp32), nwalkers = 4096 walkers evaluated per step.  One "step" is one
pass of the serving hot path over the batch: prior map -> input transform -> MLP forward ->
output transform -> Gaussian log-likelihood / T + ln prior  (== 4096 x util.Log_prob.__call__
of the reference).  Inputs are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]

For N > 1 launch with torch.distributed.run (one rank per GPU); walkers shard across ranks
with no data-path collective (weak scaling: 4096 walkers per GPU).  Rank 0 prints ONE JSON line.

Besides the contract's fields the line carries: ``roofline`` (dominant kernel, HIP-event timing, p10/p50/p90 of
single launches, PMC traffic from profiles/), ``cpu_baseline`` (N = 1), ``mcmc`` (ensemble iterations/s of the
bare sampler loop), ``strong_scaling`` (N > 1: the same 4096 walkers split over the ranks), ``training``
(optimiser steps of ChtoModelv2(26,457), batch 500 per GPU, gradient all-reduce for N > 1, with its own roofline
object) and two more serving workloads timed by HIP events on rank 0: ``chto_v2`` (the reference's network class,
ChtoModelv2(33,33)) and ``dense_1000`` (BASELINE configs[3]: ChtoModelv2(40,1000), dense inverse covariance).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

NWALKERS = 4096
NIN = NOUT = 33
WIDTH, DEPTH = 512, 4
FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
MACS_PER_EVAL = NIN * WIDTH + (DEPTH - 1) * WIDTH * WIDTH + WIDTH * NOUT     # 820 224


def _stage(msg):
    """Progress marker on stderr (LINNA_BENCH_TRACE=1): where a multi-rank run is, per rank."""
    if os.environ.get("LINNA_BENCH_TRACE", "0") == "1":
        print("[bench rank %s] %.1f %s" % (os.environ.get("RANK", "0"), time.perf_counter(), msg), file=sys.stderr, flush=True)


class _Watchdog(object):
    """N > 1 only.  The headline is timed without any data-path collective, but the sections behind it (strong scaling,
    training with its gradient all-reduce, the ensemble iterations with their walker exchange, the shutdown barrier) call
    collectives, and a collective that never completes would take the whole line with it: nothing is printed before the
    end.  Armed once the headline is known; if the remaining sections have not finished after `seconds`, rank 0 prints the
    line with what it has (and says so in "watchdog"), and every rank leaves with os._exit -- no exec, no retry -- with a
    NON-ZERO code: EXIT_HANG (3) when a section or the teardown did not finish, EXIT_RAISED (4) when one raised.  The line
    is out either way (self_launch relays it), and the launcher's exit code tells the driver that a collective hung or
    broke instead of calling the run clean."""
    EXIT_HANG, EXIT_RAISED = 3, 4


    def __init__(self, seconds, rank):
        import threading
        self.rank, self.line, self.stage, self.done = rank, None, "armed", False
        self.lock = threading.Lock()
        self.timer = threading.Timer(seconds + (0.0 if rank == 0 else 20.0), self._fire)   # rank 0 first: its line must get out
        self.timer.daemon = True
        self.seconds = seconds

    def arm(self, line):
        self.line = line
        self.timer.start()

    def emit(self, line):
        """The one place the JSON line is printed; False if the watchdog already printed it."""
        with self.lock:
            if self.done:
                return False
            self.done = True
        self.timer.cancel()
        print(json.dumps(line), file=sys.__stdout__, flush=True)
        return True

    def failed(self, exc):
        """An exception behind the headline (a collective that broke, a peer that left): the headline line still goes out
        from rank 0, the other ranks leave quietly -- nothing they could still do would change the line."""
        import traceback
        traceback.print_exc()
        with self.lock:
            if self.done:                            # the line is out already (emitted, or printed by the timer): only the code is left to give
                os._exit(self.EXIT_RAISED)
            self.done = True
        if self.rank == 0 and self.line is not None:
            self.line["watchdog"] = "a section after the headline raised %s at stage '%s'; the remaining sections are missing from this line" % (repr(exc)[:200], self.stage)
            print(json.dumps(self.line), file=sys.__stdout__, flush=True)      # (sys.stdout may be redirected by the section that hangs)
        os._exit(self.EXIT_RAISED)

    def _fire(self):
        with self.lock:
            if self.done:                            # emit() won the race: the run is finishing normally
                return
            self.done = True
        if self.rank == 0 and self.line is not None:
            self.line["watchdog"] = "sections after the headline did not finish within %.0f s (last stage: %s); they are missing from this line" % (self.seconds, self.stage)
            print(json.dumps(self.line), file=sys.__stdout__, flush=True)      # (sys.stdout may be redirected by the section that hangs)
        sys.stderr.write("[bench rank %d] watchdog: leaving at stage '%s'\n" % (self.rank, self.stage))
        sys.stderr.flush()
        os._exit(self.EXIT_HANG)


def build_problem(device):
    """README.rst:69-83 shaped problem with a random-init emulator (seed 1234, Xavier-uniform
    weights, bias 0.01 -- nn.py:97-99); synthetic: there is no trained checkpoint offline."""
    import torch
    from linna_amd import nn, util, predictor_gpu
    rs = np.random.RandomState(0)
    means = rs.uniform(size=NOUT)
    cov = np.diag(0.1 * rs.uniform(0.05, 1.0, size=NOUT))
    priors = [{"param": "test_%d" % i, "dist": "flat", "arg1": -5.0, "arg2": 5.0} for i in range(NIN)]
    torch.manual_seed(1234)
    model = nn.MLP(NIN, NOUT, None, width=WIDTH, depth=DEPTH)
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32))
    X_mean, X_std = np.zeros(NIN), np.full(NIN, 10.0 / np.sqrt(12.0))
    y_mean, y_std = means / np.sqrt(np.diag(cov)), np.ones(NOUT)
    pred = predictor_gpu.Predictor(NIN, NOUT, model=model, device=device,
                                   X_transform=util.X_transform_class(t(X_mean), t(X_std), "cpu", None),
                                   y_transform=util.Y_transform_class(t(y_mean), t(y_std), "cpu"))
    sigma = np.sqrt(np.diag(cov))
    lp = util.Log_prob(t(means), t(np.linalg.inv(cov)), pred, util.Y_invtransform_data(sigma, "cpu"),
                       util.Transform(priors), 1.0, util.gaussianlogliklihood, nograd=True)
    consts = dict(means=means, cov=cov, priors=priors, X_mean=X_mean, X_std=X_std, y_mean=y_mean, y_std=y_std,
                  sigma=sigma, weights={k: v.cpu().numpy().copy() for k, v in model.state_dict().items()})
    return lp, model, consts


def cpu_baseline(consts, z, budget_s=15.0, gpu_out=None):
    """The numpy oracle (CPU restatement of the reference path) timed on this host: batched
    BLAS evaluation (value; the faster of a 32-thread and an all-core BLAS pool) and the
    reference-faithful per-walker loop (batch 1, one thread)."""
    from oracle import likelihood
    from threadpoolctl import threadpool_limits
    emu = likelihood.Emulator("MLP", NIN, NOUT, consts["weights"], consts["X_mean"], consts["X_std"], consts["y_mean"],
                              consts["y_std"], consts["sigma"], width=WIDTH, depth=DEPTH)
    invcov = np.linalg.inv(consts["cov"])
    f = lambda zz: likelihood.log_prob(zz, emu, consts["priors"], consts["means"], invcov, 1.0)
    from linna_amd.util import cpu_quota
    cores = cpu_quota()                    # affinity capped by the container's CPU bandwidth quota (16 of 256 on the GPU box)
    best = (0.0, 0, 0)
    for nthreads in sorted({min(32, cores), cores}):
        with threadpool_limits(limits=nthreads):
            f(z)                                     # warm up the BLAS pool
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < budget_s * 0.3:
                f(z)
                n += 1
            rate = n * len(z) / (time.perf_counter() - t0)
        if rate > best[0]:
            best = (rate, nthreads, n)
    check = {}
    if gpu_out is not None:               # the GPU's lnP of the SAME walkers against the oracle's (fp32 both; a free self-check)
        ref = f(z).astype(np.float64)
        check = {"max_rel_err_vs_oracle": float(np.max(np.abs(np.asarray(gpu_out, np.float64) - ref) / np.abs(ref)))}
    with threadpool_limits(limits=1):
        m, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s * 0.3:
            f(z[m % len(z)][None, :])
            m += 1
        per_walker = m / (time.perf_counter() - t0)
    return dict({"value": best[0], "unit": "evals/s", "cores": best[1], "kind": "port",
                 "sample": "numpy oracle (oracle/likelihood.log_prob), fp32: %d passes of the same %d-walker batch with a "
                           "%d-thread BLAS pool (this process may use %d CPUs); reference-faithful per-walker loop (batch 1, one "
                           "thread): %.0f evals/s over %d calls" % (best[2], len(z), best[1], cores, per_walker, m)}, **check)


def _cpu_timed(fn, units_per_call, unit, what, budget_s=5.0):
    """`cpu_baseline` object of a secondary bench entry: the numpy oracle's restatement of the same work, timed on this
    host with the BLAS pool the CPU quota allows, on a bounded sample (`budget_s` seconds of calls)."""
    from threadpoolctl import threadpool_limits
    from linna_amd.util import cpu_quota
    cores = cpu_quota()
    with threadpool_limits(limits=cores):
        fn()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            fn()
            n += 1
        dt = time.perf_counter() - t0
    return {"value": n * units_per_call / dt, "unit": unit, "cores": cores, "kind": "port",
            "sample": "%s: %d calls in %.1f s, %d-thread BLAS pool, fp32" % (what, n, dt, cores)}


def _oracle_emulator(kind, nin, nout, model, X_mean, X_std, y_mean, y_std, sigma, **kw):
    from oracle import likelihood
    w = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
    return likelihood.Emulator(kind, nin, nout, w, X_mean, X_std, y_mean, y_std, sigma, **kw)


TRAFFIC_FILE = "r06_pmc_traffic.json"


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel.  PMC counters cannot be read from inside the timed process (rocprofv3
    wraps the program): the figure comes from the committed rocprofv3 --pmc passes of THIS round's kernel
    (tools/pmc_traffic.py: FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE, separate passes) and is labelled as
    such in the line (``traffic_source``); None if not collected."""
    try:
        with open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)) as f:
            return json.load(f)["traffic_bytes_per_launch"]
    except Exception:
        return None


def mcmc_rate(lp, nwalkers, world=1, sync=None, nsteps=1000, warm=500, exchange="none"):
    """Ensemble (stretch-move) iterations per second with every walker advanced once per iteration: 2 half steps, each
    ONE launch (proposal -> whole-network lnP of nwalkers/2 -> accept).  N > 1: EVERY rank runs this with its own
    ``nwalkers`` walkers; ``exchange="none"`` (the drivers' default, sampler._Ranks "local"): a sub-ensemble per rank, partners
    from the rank's own complementary half, no collective inside an iteration; ``"allgather"``: the partners of a half step
    from the complementary walkers of ALL ranks (one RCCL all-gather of [nwalkers/2, ndim] per half step through
    linna_allgather_f32): one ensemble of N x nwalkers walkers.  The rate is that of the slowest rank between two barriers."""
    import torch
    from linna_amd import sampler
    ens = sampler.EnsembleSampler(nwalkers, NIN, lp, seed=1, exchange=exchange if world > 1 else "none")
    ens.set_state(0.05 * np.random.RandomState(7 + ens.rank).standard_normal((nwalkers, NIN)))
    ens.run(warm, store=False)             # untimed: > 50 ms of work, past the clock ramp
    torch.cuda.synchronize()
    if sync is not None:
        sync()
    t0 = time.perf_counter()
    ens.run(nsteps, store=False)
    torch.cuda.synchronize()
    if sync is not None:
        sync()
    dt = time.perf_counter() - t0
    return dt, {"steps_per_s": nsteps / dt, "walker_updates_per_s": nsteps * nwalkers * world / dt,
                "walkers_total": nwalkers * world,
                "exchange": ("allgather of the complementary half per half step" if exchange == "allgather" else
                             "none: a sub-ensemble per rank (the drivers' default), chain gathered per convergence check") if world > 1 else "none",
                "acceptance": float(ens.naccept.float().mean()) / ens.iteration}


def driver_rate(lp, nwalkers, nsamp=2000, tmpdir=None, prefix="driver_", method="emcee"):
    """The reference's emcee driver end to end (sampler.py:458-554 -> linna_amd.sampler.HMCSampler.sample): 100 burn-in
    iterations + restart, then `nsamp` iterations with everything a run does -- chain blocks device -> host, the
    reference's HDF5 layout appended every 100 iterations (chain + chain_transformed + log_prob: 1.1 MB per iteration
    at 4096 walkers), theta of every stored sample, the integrated-autocorrelation check at EVERY 100 iterations (the
    reference's cadence; incremental on the device, csrc/autocorr.hip) -- into a directory that is removed afterwards.
    `breakdown`: the driver's own profile (host seconds / device seconds per phase; the phases overlap: sampling, the
    statistics stream and the two writer threads run concurrently)."""
    import shutil
    import tempfile
    import contextlib
    import io
    import torch
    from linna_amd import sampler, util
    priors = [{"param": "p%d" % i, "dist": "flat", "arg1": -5.0, "arg2": 5.0} for i in range(NIN)]
    x0 = 0.05 * np.random.RandomState(7).standard_normal((nwalkers, NIN))
    out = tempfile.mkdtemp(prefix="linna_bench_chain_", dir=tmpdir)
    prof = {}
    import gc
    gcs = {"t": 0.0, "n": 0, "t0": 0.0}

    def gc_cb(phase, info):                # seconds the interpreter's cyclic collector holds the driver's thread
        if phase == "start":
            gcs["t0"] = time.perf_counter()
        else:
            gcs["t"] += time.perf_counter() - gcs["t0"]; gcs["n"] += 1
    gc.callbacks.append(gc_cb)
    try:
        if method == "zeus":               # the reference's default sampler (main.py:22), its callback's cadence and 20 % discard
            drv = sampler.ZeusSampler(lp, NIN, nwalkers, x0=x0, transform=util.Transform(priors))
        else:
            drv = sampler.HMCSampler(lp, None, None, NIN, nwalkers, x0=x0, transform=util.Transform(priors))
        with contextlib.redirect_stdout(io.StringIO()):
            t0 = time.perf_counter()
            store = drv.sample(None, nsamp, outdir=out, ntimes=1e9, tautol=1e-9, incremental=True, profile=prof)   # never "converged": runs nsamp
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        n = sum(len(c) for c in store.chain)
        size = os.path.getsize(os.path.join(out, "zeus_256.h5" if method == "zeus" else "chemcee_256.h5"))
    finally:
        gc.callbacks.remove(gc_cb)
        shutil.rmtree(out, ignore_errors=True)
    prof["host_gc_s"], prof["gc_collections"] = gcs["t"], gcs["n"]
    checks = n // 100
    bd = {k: round(v, 4) if isinstance(v, float) else v for k, v in sorted(prof.items())}
    bd["checks"] = checks
    if checks and "gpu_stats_s" in prof:
        bd["stats_us_per_check"] = 1e6 * prof["gpu_stats_s"] / checks
        bd["stats_frac_of_sampling"] = prof["gpu_stats_s"] / max(prof.get("gpu_sampling_s", 0.0), 1e-12)
    burn = 0 if method == "zeus" else 100
    return {prefix + "steps_per_s": (n + burn) / dt, prefix + "iterations": n + burn, prefix + "seconds": dt,
            prefix + "chain_file_bytes": size, prefix + "tmp_fs": _fs_type(tmpdir or tempfile.gettempdir()),
            prefix + "breakdown": bd}


def _fs_type(path):
    """File-system type of the mount that holds ``path`` (/proc/mounts, longest matching mount point)."""
    best = ("", "?")
    try:
        path = os.path.realpath(path)
        with open("/proc/mounts") as f:
            for ln in f:
                parts = ln.split()
                mp = parts[1]
                if (path == mp or path.startswith(mp.rstrip("/") + "/")) and len(mp) > len(best[0]):
                    best = (mp, parts[2])
    except Exception:                                               # noqa: BLE001
        pass
    return "%s on %s" % (best[1], best[0] or "?")


def _lib_launches(model, B):
    from linna_amd import _lib
    n = _lib.load().linna_net_train_launches(model.net_handle(with_grads=True), int(B))
    return n if n > 0 else None


def training_rate(device, world, rank, backend, nsteps=150):
    """BASELINE configs[2] shape: ChtoModelv2(26, 457) (3x2pt-like stand-in), dense covariance, batch 500 PER RANK,
    one all-reduce of the flat gradient per step when N > 1 (RCCL over xGMI), lr * N (predictor_gpu.py:246).
    Full optimiser steps: gather -> forward -> chi2-ratio loss -> backward -> [all-reduce] -> AdamW."""
    nsteps = nsteps if (world == 1 or backend == "nccl") else 30          # (a gloo rehearsal stages through the host)
    import torch
    import torch.distributed as dist
    from linna_amd import nn, util, predictor_gpu, trainer
    nin, nout, B, n = 26, 457, 500, 100000             # configs[2]: "MLP training 100k samples" resident in HBM (193 MB per rank)
    rs = np.random.RandomState(5)
    q, _ = np.linalg.qr(rs.standard_normal((nout, nout)))
    cov = (q * (np.logspace(0, -2, nout) * 0.1)[None, :]) @ q.T
    cov = 0.5 * (cov + cov.T)
    data, sigma = rs.uniform(size=nout), np.sqrt(np.diag(cov))
    X_mean, X_std = rs.uniform(-0.5, 0.5, nin).astype(np.float32), rs.uniform(0.5, 3.0, nin).astype(np.float32)
    y_mean, y_std = rs.uniform(-0.5, 0.5, nout).astype(np.float32), rs.uniform(0.5, 2.0, nout).astype(np.float32)
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32))
    torch.manual_seed(1234)
    model = nn.ChtoModelv2(nin, nout, None)
    pred = predictor_gpu.Predictor(nin, nout, model=model, device=device,
                                   X_transform=util.X_transform_class(t(X_mean), t(X_std), "cpu", None),
                                   y_transform=util.Y_transform_class(t(y_mean), t(y_std), "cpu"))
    rs = np.random.RandomState(100 + rank)                           # every rank its own rows
    X = (X_mean[None, :] + X_std[None, :] * rs.standard_normal((n, nin))).astype(np.float32)
    Y = (data[None, :] + 3 * sigma[None, :] * rs.standard_normal((n, nout))).astype(np.float32)
    ytd = util.Y_transform_data(sigma, "cpu")
    yinv = util.Y_invtransform_class(t(y_mean), t(y_std), t(data), "cpu")
    lf = util.Loss_fn(t(data), torch.tensor(cov, dtype=torch.float64), torch.tensor(np.linalg.inv(cov), dtype=torch.float64), ytd, yinv, "cpu")
    loader = predictor_gpu.BatchLoader(util.ArrayDataset(X, Y), B, shuffle=True, drop_last=True)
    eng = trainer.TrainEngine(pred, loader, lf, None, world_size=world, dist_group=None)
    opt = predictor_gpu._AdamWState(model, 1e-4 * world, weight_decay=1e-4)
    perm = torch.stack(loader.epoch_batches()).to(torch.int32).to(device)
    k = [0]

    def step():
        eng.step(opt, perm[k[0] % len(perm)]); k[0] += 1
    if world == 1:
        t_end = time.perf_counter() + 0.4
        while time.perf_counter() < t_end:
            for _ in range(8):
                step()
            torch.cuda.synchronize()
    else:                                   # a step holds a collective: every rank must run the SAME number of them
        for _ in range(400 if backend == "nccl" else 20):
            step()
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nsteps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tm = torch.tensor([dt], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dt = float(tm.item())
    loss = float(eng.loss_mean.item())
    res = {"workload": "ChtoModelv2(26,457), dense covariance, resident training set of %d rows per GPU, shuffled batches of 500 per GPU, AdamW, gradient all-reduce per step for N > 1" % n,
           "resident_rows": n,
           "samples_per_s": world * B * nsteps / dt, "ms_per_step": 1e3 * dt / nsteps, "global_batch": world * B, "steps": nsteps,
           "loss_finite": bool(np.isfinite(loss)),
           # one rank: one C call, two launches (forward + loss + dX chain; parameter gradients with AdamW in the epilogue);
           # data parallel: the all-reduce sits between backward and update, AdamW is a launch of its own
           "launches_per_step": (_lib_launches(model, B) if (world == 1 and getattr(eng, "one_update", None) is True) else None)}
    # algorithmic work of one step on one rank: 3 x forward MLP FLOP per sample + the loss's 3 x 2 nout^2
    flop = B * (3 * 2.0 * model.macs_per_eval() + 6.0 * nout * nout)
    res["roofline"] = {"bound": "mfma", "flop_per_step": flop, "achieved": flop / (dt / nsteps) / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS,
                       "unit": "TFLOP/s per GPU", "frac": flop / (dt / nsteps) / 1e12 / FP32_MFMA_PEAK_TFLOPS}
    if world == 1:
        try:                               # the oracle's optimiser step (forward, loss, backward, AdamW), batch 500, all cores
            from oracle import training as otr
            stats = dict(X_mean=X_mean, X_std=X_std, y_mean=y_mean, y_std=y_std, sigma=sigma.astype(np.float32),
                         data_norm=otr.normalise_target(data[None, :], sigma, y_mean, y_std)[0],
                         icov_norm=otr.normalised_inverse_cov(cov, sigma, y_std))
            params = {k_: v_.detach().cpu().numpy().copy() for k_, v_ in model.state_dict().items()}
            ost = otr.new_opt_state(params)
            Xb, Yb = X[:B], Y[:B]
            res["cpu_baseline"] = _cpu_timed(lambda: otr.train_step(params, ost, Xb, Yb, stats, "ChtoModelv2", nin, nout, 1e-4), B, "samples/s",
                                             "numpy oracle (oracle/training.train_step: forward, chi2-ratio loss, backward, AdamW), batch %d" % B, 5.0)
        except Exception as e:                                      # noqa: BLE001
            res["cpu_baseline"] = {"error": repr(e)[:200]}
    if world > 1:
        # the gradient all-reduce alone (flat fp32 buffer + the loss scalar), HIP events on the launch stream
        from linna_amd import dist as ldist, _lib
        g = model.flat_grads()
        e0, e1, ms = C.c_void_p(), C.c_void_p(), C.c_float()
        _lib.call("linna_event_create", C.byref(e0)); _lib.call("linna_event_create", C.byref(e1))
        for _ in range(20):
            ldist.allreduce_grads(g, eng.loss_mean)
        _lib.call("linna_event_record", e0, _lib.stream())
        for _ in range(100):
            ldist.allreduce_grads(g, eng.loss_mean)
        _lib.call("linna_event_record", e1, _lib.stream())
        _lib.call("linna_event_elapsed_ms", e0, e1, C.byref(ms))
        _lib.call("linna_event_destroy", e0); _lib.call("linna_event_destroy", e1)
        res["allreduce_us"] = 1e3 * ms.value / 100
        res["allreduce_bytes"] = 4 * (g.numel() + 1)
        res["allreduce_transport"] = "RCCL through linna_allreduce_sum_f32" if ldist.comm_active(g) else "torch.distributed (%s)" % backend
    return res


def training_epochs(device, nepochs=60):
    """Whole epochs of Predictor.train (predictor_gpu.py:268-449) -- optimiser steps, the validation pass, the controller's
    record, the epoch's one host wait, controller and checkpoint bookkeeping -- at the two ends of the reference's schedule
    (main.py:22: 10 000 training rows = 20 steps of 500 in iteration 0, 40 000 = 80 steps in iteration 3; 500 / 2000
    validation rows) for ChtoModelv2(33,33) and ChtoModelv2(26,457) with a dense covariance.  `frac_in_steps` = device time
    of the optimiser steps / wall time of the epoch loop."""
    import tempfile
    import shutil
    import contextlib
    import io
    import torch
    from linna_amd import nn, util, predictor_gpu
    out = {}
    for nin, nout, dense in ((33, 33, False), (26, 457, True)):
        rs = np.random.RandomState(5)
        if dense:
            q, _ = np.linalg.qr(rs.standard_normal((nout, nout)))
            cov = (q * (np.logspace(0, -2, nout) * 0.1)[None, :]) @ q.T
            cov = 0.5 * (cov + cov.T)
        else:
            cov = np.diag(0.1 * rs.uniform(0.05, 1.0, size=nout))
        data, sigma = rs.uniform(size=nout), np.sqrt(np.diag(cov))
        X_mean, X_std = rs.uniform(-0.5, 0.5, nin).astype(np.float32), rs.uniform(0.5, 3.0, nin).astype(np.float32)
        y_mean, y_std = rs.uniform(-0.5, 0.5, nout).astype(np.float32), rs.uniform(0.5, 2.0, nout).astype(np.float32)
        t = lambda a: torch.as_tensor(np.asarray(a, np.float32))
        ytd = util.Y_transform_data(sigma, "cpu")
        yinv = util.Y_invtransform_class(t(y_mean), t(y_std), t(data), "cpu")
        largs = (t(data), torch.tensor(cov, dtype=torch.float64), torch.tensor(np.linalg.inv(cov), dtype=torch.float64), ytd, yinv, "cpu")
        lf, vf = util.Loss_fn(*largs), util.Val_metric_fn(*largs)
        for n, nv in ((10000, 500), (40000, 2000)):
            torch.manual_seed(1234)
            model = nn.ChtoModelv2(nin, nout, None)
            pred = predictor_gpu.Predictor(nin, nout, model=model, device=device, optim="automatic",
                                           X_transform=util.X_transform_class(t(X_mean), t(X_std), "cpu", None),
                                           y_transform=util.Y_transform_class(t(y_mean), t(y_std), "cpu"))
            A = rs.standard_normal((nout, nin)) * 0.3          # a learnable map: the controller takes its ordinary path (noise targets
                                                                   # would plateau and trigger the "bad training" re-initialisation every 10 epochs)
            def mk(m):
                x = rs.standard_normal((m, nin))
                return ((X_mean[None, :] + X_std[None, :] * x).astype(np.float32),
                        (data[None, :] + sigma[None, :] * (3 * np.tanh(x @ A.T) + 0.3 * rs.standard_normal((m, nout)))).astype(np.float32))
            (X, Y), (VX, VY) = mk(n), mk(nv)
            loader = predictor_gpu.BatchLoader(util.ArrayDataset(X, Y), 500, shuffle=True, drop_last=True)
            vloader = predictor_gpu.BatchLoader(util.ArrayDataset(VX, VY), nv, shuffle=False, drop_last=False)
            pred.outdir = tempfile.mkdtemp(prefix="linna_bench_train_")
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    pred.train(loader, 8, lf, vloader, vf)                       # untimed: first-use costs, and the learning-rate range
                                                                                 # test of a run without lr.npy (predictor_gpu.py:222-238)
                    pred.optim = "automatic"
                    prof = {}
                    torch.cuda.synchronize()
                    pred.train(loader, nepochs, lf, vloader, vf, profile=prof)
                lr_used = float(np.load(os.path.join(pred.outdir, "lr.npy")))
            finally:
                shutil.rmtree(pred.outdir, ignore_errors=True)
            ne = max(prof["epochs"], 1)
            med = lambda k: 1e3 * prof.get(k + "_median_s", 0.0)
            out["v2_%d_%d.steps_%d" % (nin, nout, n // 500)] = {
                "epochs": prof["epochs"], "steps_per_epoch": n // 500, "validation_rows": nv,
                "ms_per_epoch": med("epoch"), "ms_per_epoch_mean": 1e3 * prof["epoch_s"] / ne,
                "ms_steps_gpu": med("gpu_steps"), "ms_validation_gpu": med("gpu_validation"),
                "frac_in_steps": med("gpu_steps") / max(med("epoch"), 1e-9),
                "frac_in_steps_whole_run": prof.get("gpu_steps_s", 0.0) / max(prof["epoch_s"], 1e-12),
                "host_ms_per_epoch": {k[5:-9]: 1e3 * v for k, v in sorted(prof.items()) if k.startswith("host_") and k.endswith("_median_s")},
                "final_checkpoint_ms": 1e3 * prof.get("final_checkpoint_s", 0.0),
                "speculative_epochs": prof.get("speculative_epochs", 0), "speculative_epochs_undone": prof.get("speculative_epochs_undone", 0),
                "controller_actions": prof.get("controller_actions", {}), "lr": lr_used,
                "note": "medians over the epochs (host phases partition an epoch's wall time); *_mean / *_whole_run include the epochs that write a checkpoint or re-initialise"}
    return out


def secondary_serving(device, kind, nin, nout, dense, nwalkers=4096, iters=400):
    """One more serving workload in the driver's record, timed like the headline's dominant kernel (HIP events on the launch
    stream around back-to-back launches, after a clock-ramp warm-up): the reference's own network class on the README
    problem (ChtoModelv2(33,33), diagonal covariance) and BASELINE configs[3] (~1000-dim output, dense inverse
    covariance; nin = 40 is SURVEY 8d's stand-in).  FLOP per evaluation from the network's MAC count (+ 2 nout^2 for the
    dense quadratic form, 3 nout for the diagonal one), SURVEY 8d."""
    import torch
    from linna_amd import nn, util, predictor_gpu, _lib
    rs = np.random.RandomState(11)
    data = rs.uniform(size=nout)
    if dense:
        q, _ = np.linalg.qr(rs.standard_normal((nout, nout)))
        cov = (q * (np.logspace(0, -2, nout) * 0.1)[None, :]) @ q.T
        cov = 0.5 * (cov + cov.T)
    else:
        cov = np.diag(0.1 * rs.uniform(0.05, 1.0, size=nout))
    priors = [{"param": "p%d" % i, "dist": "flat", "arg1": -5.0, "arg2": 5.0} for i in range(nin)]
    torch.manual_seed(1234)
    model = getattr(nn, kind)(nin, nout, None)
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32))
    sigma = np.sqrt(np.diag(cov))
    pred = predictor_gpu.Predictor(nin, nout, model=model, device=device,
                                   X_transform=util.X_transform_class(t(np.zeros(nin)), t(np.full(nin, 10.0 / np.sqrt(12.0))), "cpu", None),
                                   y_transform=util.Y_transform_class(t(data / sigma), t(np.ones(nout)), "cpu"))
    lp = util.Log_prob(t(data), t(np.linalg.inv(cov)), pred, util.Y_invtransform_data(sigma, "cpu"), util.Transform(priors), 1.0,
                       util.gaussianlogliklihood, nograd=True)
    z = torch.as_tensor(np.random.RandomState(5).standard_normal((nwalkers, nin)).astype(np.float32), device=device)
    out = torch.empty(nwalkers, dtype=torch.float32, device=device)
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        for _ in range(32):
            lp.evaluate(z, out=out)
        torch.cuda.synchronize()
    st = _lib.stream()
    e0, e1, ms = C.c_void_p(), C.c_void_p(), C.c_float()
    _lib.call("linna_event_create", C.byref(e0)); _lib.call("linna_event_create", C.byref(e1))
    _lib.call("linna_event_record", e0, st)
    for _ in range(iters):
        lp.evaluate(z, out=out)
    _lib.call("linna_event_record", e1, st)
    _lib.call("linna_event_elapsed_ms", e0, e1, C.byref(ms))
    _lib.call("linna_event_destroy", e0); _lib.call("linna_event_destroy", e1)
    assert torch.isfinite(out).all()
    us = 1e3 * ms.value / iters
    flop_eval = 2.0 * model.macs_per_eval() + (2.0 * nout * nout + nout if dense else 3.0 * nout)
    tf = nwalkers * flop_eval / (us * 1e-6) / 1e12
    res = {"workload": "%s(%d,%d), %s inverse covariance, %d walkers, one launch per step" % (kind, nin, nout, "dense" if dense else "diagonal", nwalkers),
           "us_per_launch": us, "evals_per_s": nwalkers / (us * 1e-6), "flop_per_eval": flop_eval, "achieved": tf,
           "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_MFMA_PEAK_TFLOPS}
    if dense:
        # SURVEY 8d prices the dense quadratic form at 2 nout^2 FLOP per evaluation; the kernel takes it as |d L|^2 with the
        # lower-triangular Cholesky factor of the inverse covariance and skips the zero block of the second column pass
        # (nout > 512): `frac` is priced on the FLOP the kernel EXECUTES, `frac_priced` on the algorithmic 2 nout^2
        # (mode 1: the second column pass starts at row 512; mode 2, 16 blocks: block b of 64 columns runs the rows from 64 b on)
        tri = _lib.load().linna_dense_tri(-1) if os.environ.get("LINNA_DENSE_FACTORED", "1") != "0" else 0
        kpad = (nout + 15) // 16 * 16
        if tri == 2 and 960 < nout <= 1024:
            quad = 2.0 * sum(64 * (kpad - 64 * b) for b in range(16))
        elif tri >= 1 and nout > 512:
            quad = 2.0 * nout * nout - 2.0 * 512 * (nout - 512)
        else:
            quad = 2.0 * nout * nout
        skipped = 2.0 * nout * nout - quad
        tf_exec = nwalkers * (flop_eval - skipped) / (us * 1e-6) / 1e12
        res.update({"achieved_priced": tf, "frac_priced": tf / FP32_MFMA_PEAK_TFLOPS, "achieved": tf_exec, "frac": tf_exec / FP32_MFMA_PEAK_TFLOPS,
                    "executed_flop_per_eval": flop_eval - skipped,
                    "dense_tri_mode": tri,
                    "note": "chi^2 = |d L|^2, L = chol(Sigma^-1) lower triangular; the zero triangle is skipped block-wise (linna_dense_tri: bit-identical "
                            "to the full product); achieved / frac count the FLOP executed (zero padding of the 64-column blocks included), *_priced "
                            "SURVEY's 2 nout^2 for the quadratic form"})
    try:                                   # the oracle on the host, same walkers, and the GPU's lnP against it
        from oracle import likelihood
        emu = _oracle_emulator(kind, nin, nout, model, np.zeros(nin), np.full(nin, 10.0 / np.sqrt(12.0)), data / sigma, np.ones(nout), sigma)
        zh, icov = z.cpu().numpy(), np.linalg.inv(cov)
        f = lambda: likelihood.log_prob(zh, emu, priors, data, icov, 1.0)
        ref = f().astype(np.float64)
        res["max_rel_err_vs_oracle"] = float(np.max(np.abs(out.cpu().numpy().astype(np.float64) - ref) / np.abs(ref)))
        res["cpu_baseline"] = _cpu_timed(f, nwalkers, "evals/s", "numpy oracle (oracle/likelihood.log_prob), the same %d walkers per call" % nwalkers, 4.0)
    except Exception as e:                                          # noqa: BLE001
        res["cpu_baseline"] = {"error": repr(e)[:200]}
    return res


def _problem33(device, kind):
    """README problem (33-D Gaussian, flat priors [-5, 5], diagonal covariance) behind `kind`(33, 33), random-init weights."""
    import torch
    from linna_amd import nn, util, predictor_gpu
    rs = np.random.RandomState(11)
    data = rs.uniform(size=NOUT)
    cov = np.diag(0.1 * rs.uniform(0.05, 1.0, size=NOUT))
    sigma = np.sqrt(np.diag(cov))
    priors = [{"param": "p%d" % i, "dist": "flat", "arg1": -5.0, "arg2": 5.0} for i in range(NIN)]
    torch.manual_seed(1234)
    model = getattr(nn, kind)(NIN, NOUT, None)
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32))
    pred = predictor_gpu.Predictor(NIN, NOUT, model=model, device=device,
                                   X_transform=util.X_transform_class(t(np.zeros(NIN)), t(np.full(NIN, 10.0 / np.sqrt(12.0))), "cpu", None),
                                   y_transform=util.Y_transform_class(t(data / sigma), t(np.ones(NOUT)), "cpu"))
    lp = util.Log_prob(t(data), t(np.linalg.inv(cov)), pred, util.Y_invtransform_data(sigma, "cpu"), util.Transform(priors), 1.0,
                       util.gaussianlogliklihood, nograd=True)
    return lp, model


def production_rates(device):
    """The reference's OWN configuration: the network it hard-wires (ChtoModelv2(33,33), nn.py:59-133) at the ensemble size its
    runs use (128 walkers: 64 proposals per half step = 16 workgroups of the 4-row engine, whose time is the small-batch floor
    of the whole-network kernel).  Raw sampler rates -- the emcee stretch move and zeus' slice move -- and the
    latency of one 64-row evaluation."""
    import torch
    lp2, model = _problem33(device, "ChtoModelv2")
    z = torch.as_tensor(np.random.RandomState(5).standard_normal((64, NIN)).astype(np.float32), device=device)
    out = torch.empty(64, dtype=torch.float32, device=device)
    us = _events_us(lambda: lp2.evaluate(z, out=out), 2000)
    _, em = mcmc_rate(lp2, 128, nsteps=3000, warm=500)
    zs = slice_rate(lp2, sizes=(128,), iters=(600,))["walkers_128"]
    return {"workload": "ChtoModelv2(33,33), diagonal inverse covariance, 128 walkers (the reference's network at the reference's ensemble size)",
            "us_per_64_row_evaluation": us, "emcee_iterations_per_s": em["steps_per_s"], "emcee_acceptance": em["acceptance"],
            "zeus_iterations_per_s": zs["iterations_per_s"], "zeus_evals_per_walker_per_iteration": zs["evals_per_walker_per_iteration"],
            "zeus_path": zs["path"]}


def _events_us(fn, iters, warm_s=0.3):
    """Average time of `fn` in microseconds: HIP events on the launch stream around `iters` back-to-back calls, after a
    clock-ramp warm-up."""
    import torch
    from linna_amd import _lib
    t_end = time.perf_counter() + warm_s
    while time.perf_counter() < t_end:
        for _ in range(8):
            fn()
        torch.cuda.synchronize()
    st = _lib.stream()
    e0, e1, ms = C.c_void_p(), C.c_void_p(), C.c_float()
    _lib.call("linna_event_create", C.byref(e0)); _lib.call("linna_event_create", C.byref(e1))
    _lib.call("linna_event_record", e0, st)
    for _ in range(iters):
        fn()
    _lib.call("linna_event_record", e1, st)
    _lib.call("linna_event_elapsed_ms", e0, e1, C.byref(ms))
    _lib.call("linna_event_destroy", e0); _lib.call("linna_event_destroy", e1)
    return 1e3 * ms.value / iters


def identity_state(kind, seed=1234, shift=2.0):
    """Weights that make `kind`(33, 33) compute h(x) = x EXACTLY on |x| < shift while every other unit keeps random
    (He-scaled) weights -- the trained emulator of the README problem (theory = identity) that no offline training has to
    be waited for, with the arithmetic (and the power draw) of a real weight set.  x passes as relu(x) - relu(-x) through
    66 units of every hidden layer (identity rows there, zero rows in the layers that would mix other units in);
    ChtoModelv2's 33-unit ReLU bottleneck before its last layer (nn.py:85-86, 127-129) passes x + shift > 0 and the last
    layer's bias takes the shift off again."""
    rs = np.random.RandomState(seed)
    n = NIN
    I = np.eye(n, dtype=np.float32)
    he = lambda N, K: (np.sqrt(2.0 / K) * rs.standard_normal((N, K))).astype(np.float32)
    sd = {}

    def first(key, N):                       # [x] -> [relu(x); relu(-x); random units]
        W, b = he(N, n), (0.1 * rs.standard_normal(N)).astype(np.float32)
        W[:n], W[n:2 * n], b[:2 * n] = I, -I, 0.0
        sd[key + ".weight"], sd[key + ".bias"] = W, b

    def carry(key, N, K, bias=True):         # units 0..65 pass through, the others are random
        W = he(N, K)
        W[:2 * n] = 0.0
        W[np.arange(2 * n), np.arange(2 * n)] = 1.0
        sd[key + ".weight"] = W
        if bias:
            b = (0.1 * rs.standard_normal(N)).astype(np.float32); b[:2 * n] = 0.0
            sd[key + ".bias"] = b

    if kind == "MLP":
        first("layer1", WIDTH)
        for i in range(2, DEPTH + 1):
            carry("layer%d" % i, WIDTH, WIDTH)
        W = np.zeros((n, WIDTH), np.float32); W[:, :n], W[:, n:2 * n] = I, -I
        sd["layer%d.weight" % (DEPTH + 1)], sd["layer%d.bias" % (DEPTH + 1)] = W, np.zeros(n, np.float32)
        return sd
    assert kind == "ChtoModelv2"
    h = 1000
    first("layer1", h)
    for name, c in (("layer2", 16), ("layer3", 32), ("layer4", 64)):
        sd[name + ".layer1.weight"], sd[name + ".layer1.bias"] = he(c, h), (0.1 * rs.standard_normal(c)).astype(np.float32)
        W2, b2 = he(h // 2, c), (0.1 * rs.standard_normal(h // 2)).astype(np.float32)
        W2[:2 * n], b2[:2 * n] = 0.0, 0.0                            # 0.1 (W2 h + b2) adds nothing to the carried units
        sd[name + ".layer2.weight"], sd[name + ".layer2.bias"] = W2, b2
        carry(name + ".skip_layer", h // 2, h, bias=False)
        h //= 2
    carry("layer6", 4 * h, h)
    W7 = np.zeros((n, 4 * h), np.float32); W7[:, :n], W7[:, n:2 * n] = I, -I
    sd["layer7.weight"], sd["layer7.bias"] = W7, np.full(n, shift, np.float32)     # relu(x + shift) = x + shift
    sd["layer8.weight"], sd["layer8.bias"] = I.copy(), np.full(n, -shift, np.float32)
    return sd


def _problem33_identity(device, kind):
    """README problem (README.rst:69-83: 33-D Gaussian likelihood, flat priors [-5, 5], theory = identity) behind an
    emulator of class `kind` that IS the identity (identity_state): the posterior is the analytic one, so acceptance
    rates and autocorrelation times mean what they would in a converged LINNA iteration."""
    import torch
    from linna_amd import nn, util, predictor_gpu
    rs = np.random.RandomState(11)
    data = rs.uniform(size=NOUT)
    cov = np.diag(0.1 * rs.uniform(0.05, 1.0, size=NOUT))
    sigma = np.sqrt(np.diag(cov))
    priors = [{"param": "p%d" % i, "dist": "flat", "arg1": -5.0, "arg2": 5.0} for i in range(NIN)]
    model = getattr(nn, kind)(NIN, NOUT, None)
    model.load_state_dict(identity_state(kind))
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32))
    xs = 10.0 / np.sqrt(12.0)                 # x = theta / xs in (-1.74, 1.74); m = (h * (xs / sigma) + 0) * sigma = theta
    pred = predictor_gpu.Predictor(NIN, NOUT, model=model, device=device,
                                   X_transform=util.X_transform_class(t(np.zeros(NIN)), t(np.full(NIN, xs)), "cpu", None),
                                   y_transform=util.Y_transform_class(t(np.zeros(NOUT)), t(xs / sigma), "cpu"))
    lp = util.Log_prob(t(data), t(np.linalg.inv(cov)), pred, util.Y_invtransform_data(sigma, "cpu"), util.Transform(priors), 1.0,
                       util.gaussianlogliklihood, nograd=True)
    return lp, model, data, sigma


def hmc_rate(device, nchains=4096, nleap=5, nsamp=400):
    """BASELINE configs[4] (HMCSampler.py:19-68 batched over walkers; one process per GPU, chains are independent: no
    collective): lnP + d lnP / d z per launch (linna_logprob_grad), and whole HMC transitions of `nleap` leapfrog steps
    (momentum draw, half kick, nleap x (drift, gradient, kick), Metropolis test: 3 + nleap launches).  For the reference's
    network class ChtoModelv2(33,33) and the 4 x 512 MLP, each holding the identity-exact emulator of the README problem
    (identity_state), unit mass, the step size tuned per model to an acceptance of 0.65-0.8.
    Reported next to the kernel figure: acceptance, the integrated autocorrelation time of the chains (emcee's estimator
    on the device, averaged over chains, worst parameter) and effective samples per second = chains x samples / tau / time.
    One gradient evaluation = forward + dX-only backward = 2 x the forward FLOP."""
    import torch
    from scipy.special import erf
    from linna_amd import sampler
    out = {"chains": nchains, "leapfrog_per_sample": nleap,
           "workload": "33-D Gaussian (README), identity-exact emulator, %d independent chains, unit mass, step size tuned to 0.65-0.8 acceptance" % nchains}
    for key, kind in (("chto_v2", "ChtoModelv2"), ("mlp4x512", "MLP")):
        lp, model, data, sigma = _problem33_identity(device, kind)
        rs = np.random.RandomState(5)
        # start at the posterior: theta ~ N(data, sigma^2), z = Phi^-1((theta + 5) / 10)
        from scipy.special import ndtri
        theta0 = data[None, :] + sigma[None, :] * rs.standard_normal((nchains, NIN))
        z0 = ndtri((theta0 + 5.0) / 10.0).astype(np.float32)
        z = torch.as_tensor(z0, device=device)
        lnp = torch.empty(nchains, dtype=torch.float32, device=device)
        g = torch.empty(nchains, NIN, dtype=torch.float32, device=device)
        lp.evaluate_with_grad(z, out=lnp, grad=g)
        # the emulator is the identity: lnP equals the analytic log-posterior
        th = 10.0 * 0.5 * (1.0 + erf(z0.astype(np.float64) / np.sqrt(2.0))) - 5.0
        exact = -0.5 * (((th - data[None, :]) / sigma[None, :]) ** 2).sum(1) - 0.5 * (z0.astype(np.float64) ** 2).sum(1)
        ident_err = float(np.max(np.abs(lnp.cpu().numpy() - exact) / np.abs(exact)))
        us = _events_us(lambda: lp.evaluate_with_grad(z, out=lnp, grad=g), 300)
        assert torch.isfinite(g).all() and torch.isfinite(lnp).all()
        cpu = None
        try:                               # the oracle's lnP + gradient of the same chains on the host (oracle/likelihood.grad_log_prob)
            from oracle import likelihood
            xs_ = 10.0 / np.sqrt(12.0)
            emu = _oracle_emulator(kind, NIN, NOUT, model, np.zeros(NIN), np.full(NIN, xs_), np.zeros(NOUT), xs_ / sigma, sigma,
                                   **({"width": WIDTH, "depth": DEPTH} if kind == "MLP" else {}))
            pri = [{"param": "p%d" % i, "dist": "flat", "arg1": -5.0, "arg2": 5.0} for i in range(NIN)]
            icov = np.diag(1.0 / sigma ** 2)
            fg = lambda: likelihood.grad_log_prob(z0, emu, pri, data, icov, 1.0)
            _, gref = fg()
            cpu = _cpu_timed(fg, nchains, "gradient evals/s", "numpy oracle (oracle/likelihood.grad_log_prob), the same %d chains per call" % nchains, 4.0)
            cpu["max_grad_err_vs_oracle_rowmax"] = float(np.max(np.abs(g.cpu().numpy() - gref) / np.abs(gref).max(axis=1, keepdims=True)))
        except Exception as e:                                      # noqa: BLE001
            cpu = {"error": repr(e)[:200]}
        flop = nchains * 2.0 * (2.0 * model.macs_per_eval())
        # step size: short pilot runs, multiplicative search into the 0.65-0.8 band (same chains, same seed policy)
        h = sampler.BatchedHMC(lp, z0, seed=3)
        eps, acc, tuned = 0.05, 0.0, []
        for _ in range(14):
            a0, s0 = int(h.naccept.sum().item()), int(h.step_dev.item())
            for _ in range(15):
                h.step(nleap, eps)
            acc = (int(h.naccept.sum().item()) - a0) / float(nchains * (int(h.step_dev.item()) - s0))
            tuned.append((round(eps, 5), round(acc, 3)))
            if 0.68 <= acc <= 0.78:
                break
            eps *= (1.25 if acc > 0.78 else 0.8) if abs(acc - 0.73) < 0.2 else (1.6 if acc > 0.73 else 0.6)
        us_s = _events_us(lambda: h.step(nleap, eps), 60)
        a0, s0 = int(h.naccept.sum().item()), int(h.step_dev.item())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        chain, _ = h.sample(nsamp, nleap, eps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        acc = (int(h.naccept.sum().item()) - a0) / float(nchains * (int(h.step_dev.item()) - s0))
        dc = sampler.DeviceChain()
        dc.append(chain)
        tau = dc.integrated_time()
        tau_max, tau_med = float(np.nanmax(tau)), float(np.nanmedian(tau))
        zs = chain[nsamp // 4:].reshape(-1, NIN).double()
        thm = (10.0 * 0.5 * (1.0 + torch.erf(zs / np.sqrt(2.0))) - 5.0).mean(0).cpu().numpy()
        out[key] = {"us_per_gradient_eval": us, "gradient_evals_per_s": nchains / (us * 1e-6), "achieved": flop / (us * 1e-6) / 1e12,
                    "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flop / (us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                    "flop_per_gradient_eval": flop / nchains, "us_per_hmc_sample": us_s, "launches_per_hmc_sample": 3 + nleap,
                    "leapfrog_steps_per_s": nleap * nchains / (us_s * 1e-6), "hmc_iterations_per_s": 1.0 / (us_s * 1e-6),
                    "step_size": eps, "step_size_search": tuned, "acceptance": acc, "samples_per_chain": nsamp,
                    "tau_worst_parameter": tau_max, "tau_median_parameter": tau_med,
                    "ess_per_s": nchains * nsamp / tau_max / dt, "ess_per_s_per_chain": nsamp / tau_max / dt,
                    "emulator_identity_max_rel_err": ident_err,
                    "posterior_mean_max_abs_dev_sigma": float(np.max(np.abs(thm - data) / sigma)),
                    "cpu_baseline": cpu}
    return out


def slice_rate(lp, sizes=(4096, 128), iters=(100, 600)):
    """The reference's DEFAULT sampler (main.py:22 method="zeus"; sampler.py:699-737): ensemble slice sampling iterations
    per second on the headline problem, at the bench's 4096 walkers and at the reference's own ensemble size (cosmolike:
    128 walkers), with the evaluations one iteration costs (stepping out + shrinking, per walker)."""
    import torch
    from linna_amd import sampler, _lib
    out = {}
    for nw, n in zip(sizes, iters):
        ens = sampler.SliceEnsembleSampler(nw, NIN, lp, seed=1)
        ens.set_state(0.05 * np.random.RandomState(7).standard_normal((nw, NIN)))
        ens.run(max(80, n // 4), store=False)                # tunes mu (zeus' first iterations), ramps the clocks; the later rounds' engines settle (iteration 64)
        torch.cuda.synchronize()
        e0, it0 = ens.neval, ens.iteration
        t0 = time.perf_counter()
        ens.run(n, store=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["walkers_%d" % nw] = {"path": "one C call per half step (linna_slice_half_step)" if ens._fast_ok else "round loop",
                                  "iterations_per_s": n / dt, "us_per_iteration": 1e6 * dt / n,
                                  "evals_per_walker_per_iteration": (ens.neval - e0) / max(1, ens.iteration - it0) / nw,
                                  "walker_updates_per_s": n * nw / dt, "mu": float(ens.mu),
                                  "ends_per_side_by_round": ens.m_sched, "trials_by_round": ens.nt_sched,
                                  # mean fraction of a half ensemble still active behind each round: what the look-ahead is used for
                                  "trials_used_by_round": ens.round_usage(),
                                  # the engine of a later round's launch follows the points the counters let it expect (2^20: never ran)
                                  "expected_points_by_round": ens.expected_rows, "fusion_mask": _lib.slice_fusion(-1)}
    return out


def notebook_2d(device):
    """The ONE configuration the reference publishes a rate for (docs/notebooks/multivariate_gaussian_distribution.ipynb:
    105,135,165,195 -- tqdm of its four iterations: 34.61 / 31.54 / 34.10 / 33.66 it/s; hardware not stated, the paths are
    the author's cluster): zeus, 4 walkers, ``ChtoModelv2(2,2)``, 2-D Gaussian, CPU, pool=None.  Here: the reference's own
    fixture checkpoint of exactly that model (tests/golden/2dgaussian_Fulltconn = the reference's tests/test_data) through
    ``ZeusSampler`` / ``HMCSampler`` (everything a run does: chain file in the reference's HDF5 layout, theta of every sample,
    convergence statistics every 100 iterations), and the numpy oracle's per-walker ``Log_prob`` loop on one host core priced
    at the evaluations zeus needed per iteration.  At 4 walkers a half step is two proposals: the GPU runs at its launch
    floor, the point of the figure is what a user of the notebook sees."""
    import shutil
    import tempfile
    import contextlib
    import io
    import torch
    from linna_amd import sampler, util, nn
    fix = os.path.join(ROOT, "tests", "golden", "2dgaussian_Fulltconn", "iter_0/")
    model, yinv = util.retrieve_model(fix, 2, 2, nn.ChtoModelv2)
    priors = [{"param": "test_%d" % i, "dist": "flat", "arg1": -2.0, "arg2": 2.0} for i in range(2)]
    data, cov = np.array([0.1, 1.0]), np.diag([0.5, 0.2])
    lp = util.Log_prob(data, np.linalg.inv(cov), model, yinv, util.Transform(priors), 1.0, util.gaussianlogliklihood, nograd=True)
    nw, out = 4, {"published_it_per_s": [34.61, 31.54, 34.10, 33.66],
                  "published_source": "docs/notebooks/multivariate_gaussian_distribution.ipynb:105,135,165,195 (zeus, 4 walkers, CPU; hardware unspecified)"}
    x0 = np.zeros((nw, 2)) + 1e-3 * np.random.RandomState(0).standard_normal((nw, 2))
    for method, nsamp in (("zeus", 3000), ("emcee", 6000)):
        tmp = tempfile.mkdtemp(prefix="linna_bench_nb_")
        try:
            mk = lambda: (sampler.ZeusSampler(lp, 2, nw, x0=x0, transform=util.Transform(priors)) if method == "zeus"
                          else sampler.HMCSampler(lp, None, None, 2, nw, x0=x0, transform=util.Transform(priors)))
            with contextlib.redirect_stdout(io.StringIO()):
                mk().sample(None, 300, outdir=tmp, ntimes=1e9, tautol=1e-9, overwrite=True)     # untimed: first-use costs
                drv = mk()
                t0 = time.perf_counter()
                store = drv.sample(None, nsamp, outdir=tmp, ntimes=1e9, tautol=1e-9, overwrite=True)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            n = sum(len(c) for c in store.chain) + (100 if method == "emcee" else 0)
            th = np.concatenate([np.asarray(c) for c in store.chain_transformed])
            th = th[len(th) // 4:]
            rec = {"it_per_s": n / dt, "iterations": n, "seconds": dt, "vs_published_median": n / dt / 33.88,
                   "posterior_mean": th.reshape(-1, 2).mean(0).tolist(), "posterior_std": th.reshape(-1, 2).std(0).tolist()}
            if method == "zeus" and drv.sampler is not None:
                rec["evals_per_walker_per_iteration"] = drv.sampler.neval / max(1, drv.sampler.iteration) / nw
                rec["mu"] = drv.sampler.mu
            out[method] = rec
        except Exception as e:                                      # noqa: BLE001
            out[method] = {"error": repr(e)[:300]}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    try:
        from oracle import likelihood
        w = {k: v.detach().cpu().numpy().copy() for k, v in model.model.state_dict().items()}
        g = np.load(os.path.join(ROOT, "tests", "golden", "fixture2d.npz"))
        emu = likelihood.Emulator("ChtoModelv2", 2, 2, w, g["X_mean"], g["X_std"], g["y_mean"], g["y_std"], g["sigma"])
        z = np.random.RandomState(1).standard_normal((64, 2)).astype(np.float32) * 0.5
        ic = np.linalg.inv(cov).astype(np.float32)
        f = lambda: likelihood.log_prob_per_walker(z, emu, priors, data.astype(np.float32), ic, 1.0)
        ref = f()
        got = lp(z, returntorch=False)
        cb = _cpu_timed(f, len(z), "evals/s", "oracle per-walker Log_prob loop, ChtoModelv2(2,2) fixture, one core", budget_s=3.0)
        epi = out.get("zeus", {}).get("evals_per_walker_per_iteration")
        cb["max_rel_err_vs_oracle"] = float(np.max(np.abs(got - ref) / np.abs(ref)))
        if epi:
            cb["zeus_it_per_s_at_that_rate"] = cb["value"] / (nw * epi)
        out["cpu_baseline"] = cb
    except Exception as e:                                          # noqa: BLE001
        out["cpu_baseline"] = {"error": repr(e)[:300]}
    return out


def e2e(device, nwalkers=128, nepoch=200, ntrain=10000, nval=500):
    """Where the wall time of a whole ``ml_sampler_core`` run goes (main.py:139-334) once the kernels are fast: the README
    problem (33-D Gaussian, identity theory, flat priors), the network the reference hard-wires (ChtoModelv2(33,33)), emcee,
    128 walkers, the reference's 4-iteration schedule (10 000 training + 500 validation points per iteration, temperatures
    4, 2, 1, 1) with `nepoch` epochs per iteration instead of the reference's cap of 4500 (its early stopping ends most runs
    far below the cap; stated in the object) -- seconds per stage from the product's own
    stage marks (linna_amd._lib.stage): design of the training points, the user's theory() calls, text I/O of the samples,
    loading them back, the LR range test, the epochs, checkpoints, model retrieval, burn-in + sampling + statistics + chain
    file (the driver's own profile), chain read-back."""
    import shutil
    import tempfile
    import contextlib
    import io
    import torch
    from linna_amd import main as lmain, nn, _lib
    rs = np.random.RandomState(0)
    ndim = NIN
    means = rs.uniform(size=ndim)
    cov = np.diag(0.1 * rs.uniform(0.2, 1.0, size=ndim))
    init = rs.uniform(size=ndim)
    priors = [{"param": "p%d" % i, "dist": "flat", "arg1": -5.0, "arg2": 5.0} for i in range(ndim)]
    tmp = tempfile.mkdtemp(prefix="linna_bench_e2e_")
    prof = {}
    n_theory = [0]

    def theory(x, outdir):
        n_theory[0] += 1
        return x[1]
    try:
        with contextlib.redirect_stdout(io.StringIO()), _lib.stage_profile(prof):
            t0 = time.perf_counter()
            chain, lps = lmain.ml_sampler_core(
                [ntrain] * 4, [nval] * 4, [2, 2, 5, 4], [5, 5, 10, 15], [0.03, 0.03, 0.02, 0.01], [0.2] * 4, [0.15] * 4, tmp + "/", theory,
                priors, means, cov, init, None, nwalkers, "cuda", None, False, [4.0, 2.0, 1.0, 1.0], nnmodel_in=nn.ChtoModelv2,
                params={"trainingoption": 1, "num_epochs": nepoch, "batch_size": 500}, method="emcee")
            torch.cuda.synchronize()
            total = time.perf_counter() - t0
        its = []
        for i in range(4):
            from linna_amd import sampler
            d = sampler.ChainStore.load(os.path.join(tmp, "iter_%d" % i, "chemcee_256"))
            its.append(int(d["chain"].shape[0]))
        sig = np.sqrt(np.diag(cov))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    top = {k: v for k, v in prof.items() if "." not in k}
    sub = {k: round(v, 4) for k, v in sorted(prof.items()) if "." in k}
    other = total - sum(top.values())
    return {"workload": "ml_sampler_core, README 33-D Gaussian, ChtoModelv2(33,33), emcee, %d walkers, 4 iterations x (%d train + %d val "
                        "points, %d epochs of batch 500) [the reference's schedule: 10000 + 500 points, up to 4500 epochs]" % (nwalkers, ntrain, nval, nepoch),
            "total_s": total, "stage_s": {k: round(v, 4) for k, v in sorted(top.items(), key=lambda kv: -kv[1])},
            "stage_frac": {k: round(v / total, 4) for k, v in sorted(top.items(), key=lambda kv: -kv[1])},
            "substage_s": sub, "unaccounted_s": round(other, 4), "theory_calls": n_theory[0], "sampling_iterations_by_iteration": its,
            "posterior_mean_dev_sigma_max": float(np.max(np.abs(chain.mean(0) - means) / sig)),
            "posterior_std_ratio_minmax": [float(np.min(chain.std(0) / sig)), float(np.max(chain.std(0) / sig))],
            "samples_returned": int(len(chain))}


def rate_vs_walkers(lp, sizes=(64, 128, 256, 512, 1024, 2048, 4096), nsteps=600):
    """Stretch-move iterations per second against the ensemble size on the headline problem: what an ensemble costs.  Up to
    2048 walkers a half step (<= 1024 proposals) is ONE workgroup's time for the whole network (the small-batch floor): the
    iteration rate is flat and walker updates per second grow with the ensemble -- walkers are free up to there."""
    import torch
    from linna_amd import sampler
    out = {}
    for nw in sizes:
        ens = sampler.EnsembleSampler(nw, NIN, lp, seed=1)
        ens.set_state(0.05 * np.random.RandomState(7).standard_normal((nw, NIN)))
        ens.run(200, store=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ens.run(nsteps, store=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[str(nw)] = {"it_per_s": nsteps / dt, "us_per_half_step": 0.5e6 * dt / nsteps, "walker_updates_per_s": nsteps * nw / dt}
    return out


def summary(res):
    """The secondary figures in one compact object, LAST in the line (the driver keeps the line's tail): microseconds and
    fraction of the fp32-MFMA peak per workload, sampler rates, the end-to-end total."""
    g = lambda d, *ks: (g(d.get(ks[0]), *ks[1:]) if len(ks) > 1 else d.get(ks[0])) if isinstance(d, dict) else None
    r = lambda v, n=4: None if v is None else round(float(v), n)
    return {
        "headline_us": r(1e3 * res["ms_per_step"], 2), "headline_frac": r(g(res, "roofline", "frac")),
        "training_ms_per_step": r(g(res, "training", "ms_per_step")), "training_frac": r(g(res, "training", "roofline", "frac")),
        "chto_v2_us": r(g(res, "chto_v2", "us_per_launch"), 2), "chto_v2_frac": r(g(res, "chto_v2", "frac")),
        "dense_1000_us": r(g(res, "dense_1000", "us_per_launch"), 2), "dense_1000_frac": r(g(res, "dense_1000", "frac")),
        "hmc_v2_us_per_grad": r(g(res, "hmc", "chto_v2", "us_per_gradient_eval"), 2), "hmc_v2_frac": r(g(res, "hmc", "chto_v2", "frac")),
        "hmc_mlp_us_per_grad": r(g(res, "hmc", "mlp4x512", "us_per_gradient_eval"), 2), "hmc_mlp_frac": r(g(res, "hmc", "mlp4x512", "frac")),
        "mcmc_4096_it_s": r(g(res, "mcmc", "steps_per_s"), 1), "mcmc_128_it_s": r(g(res, "mcmc", "walkers_128", "steps_per_s"), 1),
        "mcmc_128_driver_it_s": r(g(res, "mcmc", "walkers_128", "driver_steps_per_s"), 1),
        "slice_4096_it_s": r(g(res, "slice", "walkers_4096", "iterations_per_s"), 1), "slice_128_it_s": r(g(res, "slice", "walkers_128", "iterations_per_s"), 1),
        "slice_128_evals_per_walker": r(g(res, "slice", "walkers_128", "evals_per_walker_per_iteration"), 2),
        "production_128_us_per_eval": r(g(res, "production_128", "us_per_64_row_evaluation"), 2),
        "production_128_emcee_it_s": r(g(res, "production_128", "emcee_iterations_per_s"), 1), "production_128_zeus_it_s": r(g(res, "production_128", "zeus_iterations_per_s"), 1),
        "notebook_2d_zeus_it_s": r(g(res, "notebook_2d", "zeus", "it_per_s"), 1), "notebook_2d_emcee_it_s": r(g(res, "notebook_2d", "emcee", "it_per_s"), 1),
        "notebook_2d_published_it_s": 33.9, "e2e_total_s": r(g(res, "e2e", "total_s"), 2),
        "e2e_largest_stage": (max(res["e2e"]["stage_s"].items(), key=lambda kv: kv[1]) if g(res, "e2e", "stage_s") else None),
        "cpu_baseline_evals_s": r(g(res, "cpu_baseline", "value"), 0),
    }


def time_dominant_kernel(lp, z, out, iters):
    """HIP-event timing (events recorded on the launch stream) of the dominant kernel: the
    whole-network serving kernel net_stream_kernel<6, 0, false, 0, 16> -- ONE launch per step evaluates prior map,
    5 layers and the log-likelihood for all walkers.  Returns (avg ms per launch, algorithmic
    FLOP per launch = nwalkers x (2 x 820 224 MACs + 99 log-likelihood FLOP))."""
    from linna_amd import _lib
    st = _lib.stream()
    for _ in range(300):                   # untimed: the host work since the timed loop let the clocks drop again
        lp.evaluate(z, out=out)
    e0, e1 = C.c_void_p(), C.c_void_p()
    _lib.call("linna_event_create", C.byref(e0)); _lib.call("linna_event_create", C.byref(e1))
    _lib.call("linna_event_record", e0, st)
    for _ in range(iters):
        lp.evaluate(z, out=out)
    _lib.call("linna_event_record", e1, st)
    ms = C.c_float()
    _lib.call("linna_event_elapsed_ms", e0, e1, C.byref(ms))
    ms_avg = ms.value / iters
    _lib.call("linna_event_destroy", e0); _lib.call("linna_event_destroy", e1)
    # spread of single launches (median and p10 / p90): one event pair per launch, 200 launches
    evs = []
    for _ in range(201):
        e = C.c_void_p(); _lib.call("linna_event_create", C.byref(e)); evs.append(e)
    _lib.call("linna_event_record", evs[0], st)
    for i in range(200):
        lp.evaluate(z, out=out)
        _lib.call("linna_event_record", evs[i + 1], st)
    per = []
    for i in range(200):
        _lib.call("linna_event_elapsed_ms", evs[i], evs[i + 1], C.byref(ms)); per.append(ms.value)
    for e in evs:
        _lib.call("linna_event_destroy", e)
    q = np.percentile(per, [10, 50, 90])
    return ms_avg, float(z.shape[0]) * (2.0 * MACS_PER_EVAL + 3 * NOUT), [float(v) for v in q]


def self_launch(ngpus, argv):
    """`python bench.py --gpus N` from a bare shell: N ranks through torch.distributed.run as a CHILD process (this
    process has not touched a GPU and never will), rendezvous on 127.0.0.1 at a free port.  Rank 0's JSON line is the
    only thing relayed to stdout; the ranks' stderr passes through.  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        if out.lstrip().startswith("{"):
            line = out.rstrip("\n")
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 1                                              # the ranks ended without a result line
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--graph", action="store_true", help="replay the step as a hipGraph instead of direct launches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-training", action="store_true", help="skip the secondary training-throughput measurement")
    ap.add_argument("--no-secondary", action="store_true", help="skip the chto_v2 / dense_1000 serving objects (they run the "
                    "same kernel instantiation as the headline: profile the headline without them)")
    ap.add_argument("--no-driver", action="store_true", help="skip mcmc.driver_steps_per_s (a 2.3 GB chain file in the temporary directory)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                                                      "rehearse the multi-rank path on a box with fewer GPUs)")
    ap.add_argument("--only", default="", help="comma-separated secondary objects to run alone (notebook_2d, e2e, slice, production_128, "
                    "rate_vs_walkers, hmc, chto_v2, dense_1000): one JSON object with just those, no headline")
    ap.add_argument("--launch-check", action="store_true", help="ranks only rendezvous (linna_amd.dist.init), sum their ranks over "
                    "the process group and rank 0 prints that: what the CPU test of the self-launcher runs (no GPU needed)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1:
        # a bare `python bench.py --gpus N`: start the N ranks as fresh child processes BEFORE anything here touches a GPU
        # (never exec from a process that has initialised HIP), relay rank 0's JSON line, exit with the launcher's code
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    if args.launch_check:
        from linna_amd import dist as ldist
        w = ldist.init(backend=args.backend if args.backend != "nccl" or torch.cuda.device_count() else "gloo", comm=False)
        if os.environ.get("LINNA_BENCH_FAIL_RANK") == os.environ.get("RANK", "0"):
            sys.exit(3)                                     # (the test of "a failing rank fails the launcher")
        t = torch.tensor([ldist.rank() + 1.0])
        ldist.allreduce_grads(t)
        if ldist.rank() == 0:
            print(json.dumps({"launch_check": True, "world": w, "rank_sum": float(t.item()), "collectives": ldist.collectives()}), flush=True)
        ldist.shutdown()
        return
    from linna_amd import _lib
    from linna_amd.util import limit_threads_to_quota
    limit_threads_to_quota()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        sys.exit("bench.py --gpus %d was launched with WORLD_SIZE=%d" % (args.gpus, world))
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and world > 1 and local_rank >= ndev:
        sys.exit("rank %d has no GPU (%d visible): one rank per GPU" % (local_rank, ndev))
    dev_index = local_rank % max(ndev, 1)           # (gloo rehearsal may share a GPU between ranks)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    collectives, comm_ranks = None, 0
    if world > 1:
        # rendezvous, then the library's own RCCL communicator behind the C ABI (bounded bring-up, self-test all-reduce,
        # MIN-agreement over the ranks, fall back to torch.distributed as the transport): linna_amd.dist.init()
        from linna_amd import dist as ldist
        ldist.init(backend=args.backend, device=device)
        collectives = ldist.collectives()
        comm_ranks = ldist.comm_info(dev_index)[1]

    _stage("process group up: %s" % collectives)
    lp, model, consts = build_problem(device)
    if args.only:
        fns = {"notebook_2d": lambda: notebook_2d(device), "e2e": lambda: e2e(device), "slice": lambda: slice_rate(lp),
               "production_128": lambda: production_rates(device), "rate_vs_walkers": lambda: rate_vs_walkers(lp), "hmc": lambda: hmc_rate(device),
               "chto_v2": lambda: secondary_serving(device, "ChtoModelv2", 33, 33, False),
               "dense_1000": lambda: secondary_serving(device, "ChtoModelv2", 40, 1000, True),
               "training": lambda: training_rate(device, 1, 0, args.backend)}
        print(json.dumps({k: fns[k]() for k in args.only.split(",")}), flush=True)
        return
    z_host = np.random.RandomState(100 + rank).standard_normal((NWALKERS, NIN)).astype(np.float32)
    z = torch.as_tensor(z_host, device=device)
    out = torch.empty(NWALKERS, dtype=torch.float32, device=device)
    st = _lib.stream()

    def step_direct():
        lp.evaluate(z, out=out)

    step_direct()
    torch.cuda.synchronize()
    graph = None
    if args.graph:
        # capture one step on a side stream (hipGraph), replay it on the same stream
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            st_side = _lib.stream()
            lp.evaluate(z, out=out)                 # allocate workspaces outside capture
            s.synchronize()
            _lib.call("linna_graph_begin", st_side)
            lp.evaluate(z, out=out)
            g = C.c_void_p()
            _lib.call("linna_graph_end", st_side, C.byref(g))
        graph = g

    def step():
        if graph is not None:
            _lib.call("linna_graph_launch", graph, st)
        else:
            step_direct()

    # Untimed: bring the GPU out of its idle clocks first.  After the host-side set-up above the chip sits at idle
    # DVFS state and the first ~50 ms of work run 2-3x slow (tools/grad_timing.py); W = 30 warm-up steps are 2 ms.
    t_ramp = time.perf_counter() + 0.5
    while time.perf_counter() < t_ramp:
        for _ in range(64):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    _stage("headline timed")
    # sanity: the timed path produced finite numbers
    assert torch.isfinite(out).all(), "non-finite log-probabilities in the timed path"

    def headline_line():
        ms_kernel, flop_launch, ms_q = time_dominant_kernel(lp, z, out, max(500, args.steps))
        achieved = flop_launch / (ms_kernel * 1e-3) / 1e12
        return {
            "metric": "emulator log-likelihood evals/sec",
            "value": world * NWALKERS * args.steps / elapsed,
            "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "33-D Gaussian, 4x512 MLP emulator (33->512x4->33), nwalkers=4096 batched "
                                   "log-likelihood per GPU (BASELINE configs[1])",
                       "nwalkers_per_gpu": NWALKERS, "flop_per_eval": 2 * MACS_PER_EVAL + 3 * NOUT,
                       "launch": "hipGraph replay" if graph is not None else "direct launches",
                       "parallelism": "walkers sharded, %d rank(s), no data-path collective" % world},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": pmc_traffic(),
                         "traffic_source": "profiles/%s (rocprofv3 --pmc passes of this kernel, not measured in this run)" % TRAFFIC_FILE,
                         "kernel": "net_stream_kernel<6, 0, false, 0, 16> (whole network per launch: 16 walkers/workgroup, activations in LDS, fragment-order weight stream loaded straight into the MFMA operand registers, v_mfma_f32_16x16x4_f32)",
                         "avg_launch_ms": ms_kernel, "launch_ms_p10_p50_p90": ms_q, "flop_per_launch": flop_launch},
            "step_tflops": world * NWALKERS * args.steps * (2 * MACS_PER_EVAL) / elapsed / 1e12,
        }

    # N > 1: from here on the sections call collectives; the headline line is put together first and a watchdog holds it
    dog = None
    res = headline_line() if rank == 0 else None
    if world > 1:
        global _DOG
        dog = _DOG = _Watchdog(float(os.environ.get("LINNA_BENCH_WATCHDOG_S", "420")), rank)
        dog.arm(res)
    def _at(stage):
        _stage(stage)
        if dog is not None:
            dog.stage = stage

    # N > 1: also the strong-scaling figure (the SAME 4096 walkers split over the ranks, BASELINE's "nwalkers=4096
    # ... at 8xMI355X"): each rank evaluates 4096 / N walkers per step on the small-batch engine of the kernel
    strong = None
    if world > 1 and NWALKERS % world == 0:
        zs, outs = z[:NWALKERS // world].contiguous(), out[:NWALKERS // world]
        for _ in range(200):
            lp.evaluate(zs, out=outs)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0s = time.perf_counter()
        for _ in range(args.steps):
            lp.evaluate(zs, out=outs)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        ts = torch.tensor([time.perf_counter() - t0s], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        strong = {"nwalkers_total": NWALKERS, "nwalkers_per_gpu": NWALKERS // world, "evals_per_s": NWALKERS * args.steps / float(ts.item()),
                  "ms_per_step": 1e3 * float(ts.item()) / args.steps}

    # secondary figure, every rank takes part (collective inside): training throughput on the configs[2] shape.
    # Guarded: a failure here must not cost the headline line.  (N > 1: every finished section goes into the held line at once --
    # the watchdog prints that dict, so a section that hangs later costs only itself and what follows it.)
    if rank == 0 and strong is not None:
        res["strong_scaling"] = strong
    _at("strong scaling done")
    training = None
    if not args.no_training:
        try:
            training = training_rate(device, world, rank, args.backend)
        except Exception as e:                                      # noqa: BLE001
            training = {"error": repr(e)[:300]}

    if training is not None and world == 1 and not args.no_training and not args.no_secondary and "error" not in training:
        try:
            training["epoch"] = training_epochs(device)
        except Exception as e:                                      # noqa: BLE001
            training["epoch"] = {"error": repr(e)[:300]}
    if rank == 0 and training is not None:
        res["training"] = training
    _at("training done: %s" % (training,))
    # ensemble iterations: every rank takes part (cross-rank partner exchange for N > 1)
    mcmc = None
    try:
        sync = (lambda: (dist.barrier(), torch.cuda.synchronize())) if world > 1 else None
        quick = world > 1 and args.backend != "nccl"                 # (a gloo rehearsal stages every all-gather through the host)
        def slowest(dt_m, m):
            if world > 1:
                tm = torch.tensor([dt_m], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                k = dt_m / float(tm.item())
                m["steps_per_s"] *= k
                m["walker_updates_per_s"] *= k
            return m
        mcmc = slowest(*mcmc_rate(lp, NWALKERS, world, sync, 100 if quick else 1000, 20 if quick else 500))
        if world > 1:
            if rank == 0:
                res["mcmc"] = mcmc                                   # (held by the watchdog from here on)
            _at("mcmc (sub-ensembles) done; one-ensemble all-gather leg")
            # the one-ensemble mode (partners from every rank: an all-gather per half step) beside the default: DESIGN section 6
            # predicts it SLOWER than one GPU -- the driver's N = 2 / 4 / 8 runs measure that prediction.  The first data-path
            # all-gather of a run with more than one rank: behind everything else that needs every rank.
            mcmc["one_ensemble_allgather"] = slowest(*mcmc_rate(lp, NWALKERS, world, sync, 100 if quick else 1000, 20 if quick else 500, "allgather"))
    except Exception as e:                                          # noqa: BLE001
        mcmc = {"error": repr(e)[:300]}

    if world == 1 and isinstance(mcmc, dict) and "error" not in mcmc:
        # the stretch move at the reference's production ensemble size (cosmolike_run.py:184: 128 walkers; README: 4)
        try:
            nw = 128
            _, m128 = mcmc_rate(lp, nw, 1, None, 4000, 1000)
            ncu = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
            half = nw // 2
            m128 = {"steps_per_s": m128["steps_per_s"], "us_per_half_step": 0.5e6 / m128["steps_per_s"],
                    "walker_updates_per_s": m128["walker_updates_per_s"], "acceptance": m128["acceptance"],
                    "proposals_per_half_step": half, "workgroups_per_half_step": (half + 3) // 4,
                    "engine_rows": 4 if half <= 4 * ncu else 8 if half <= 8 * ncu else 16,
                    "note": "one launch per half step (proposal + lnP + Metropolis test); a launch of <= 1024 rows lasts as long as ONE 4-row workgroup needs for the whole network"}
            mcmc["walkers_128"] = m128
        except Exception as e:                                      # noqa: BLE001
            mcmc["walkers_128"] = {"error": repr(e)[:300]}
    _at("mcmc done: %s" % (mcmc,))
    if rank == 0:
        if strong is not None:
            res["strong_scaling"] = strong
        if training is not None:
            res["training"] = training
        res["mcmc"] = mcmc
        if world == 1 and not args.no_driver and isinstance(mcmc, dict) and "error" not in mcmc:
            try:
                driver_rate(lp, NWALKERS, nsamp=200)               # untimed: first-use costs of a process (pinned buffers, threads, the 8-row engine's stream)
                res["mcmc"].update(driver_rate(lp, NWALKERS))
                if os.path.isdir("/dev/shm"):                      # the same run with the chain file in memory: pipeline cost without the disk
                    res["mcmc"].update(driver_rate(lp, NWALKERS, tmpdir="/dev/shm", prefix="driver_shm_"))
            except Exception as e:                                  # noqa: BLE001
                res["mcmc"]["driver_error"] = repr(e)[:300]
            if isinstance(mcmc.get("walkers_128"), dict) and "error" not in mcmc["walkers_128"]:
                try:                                                # the reference's production ensemble, a chain as long as its runs
                    mcmc["walkers_128"].update(driver_rate(lp, 128, nsamp=20000))
                except Exception as e:                              # noqa: BLE001
                    mcmc["walkers_128"]["driver_error"] = repr(e)[:300]
        if collectives is not None:
            res["collectives"] = collectives
            res["rccl_ranks"] = comm_ranks          # linna_comm_info: ranks of the library's communicator (0 = torch.distributed carries the data path)
        if not args.no_secondary:
            for key, spec in (("chto_v2", ("ChtoModelv2", 33, 33, False)), ("dense_1000", ("ChtoModelv2", 40, 1000, True))):
                _at("secondary: " + key)
                try:
                    res[key] = secondary_serving(device, *spec)
                except Exception as e:                              # noqa: BLE001
                    res[key] = {"error": repr(e)[:300]}
            for key, fn in (("hmc", lambda: hmc_rate(device)), ("slice", lambda: slice_rate(lp)),   # configs[4]; the default sampler
                            ("production_128", lambda: production_rates(device))):
                _at("secondary: " + key)
                try:
                    res[key] = fn()
                except Exception as e:                              # noqa: BLE001
                    res[key] = {"error": repr(e)[:300]}
            if world == 1:
                _at("secondary: rate_vs_walkers")
                try:
                    if isinstance(res.get("mcmc"), dict):
                        res["mcmc"]["rate_vs_walkers"] = rate_vs_walkers(lp)
                except Exception as e:                              # noqa: BLE001
                    res["mcmc"]["rate_vs_walkers"] = {"error": repr(e)[:300]}
                for key, fn in (("notebook_2d", lambda: notebook_2d(device)), ("e2e", lambda: e2e(device))):
                    if key == "e2e" and args.no_driver:
                        continue
                    _at("secondary: " + key)
                    try:
                        res[key] = fn()
                    except Exception as e:                          # noqa: BLE001
                        res[key] = {"error": repr(e)[:300]}
            if world == 1 and not args.no_driver and isinstance(res.get("slice", {}).get("walkers_128"), dict):   # (the drivers shard over the default process group: an N = 1 measurement)
                _at("secondary: zeus driver")
                try:                                                # the zeus driver end to end at the reference's ensemble size
                    res["slice"]["walkers_128"].update(driver_rate(lp, 128, nsamp=3000, method="zeus"))
                except Exception as e:                              # noqa: BLE001
                    res["slice"]["walkers_128"]["driver_error"] = repr(e)[:300]
        if not args.no_cpu_baseline and world == 1:          # the CPU leg is an N = 1 measurement
            lp.evaluate(z, out=out)
            res["cpu_baseline"] = cpu_baseline(consts, z_host, gpu_out=out.cpu().numpy())
        try:
            res["summary"] = summary(res)            # LAST: the driver keeps the tail of the line
        except Exception as e:                      # noqa: BLE001
            res["summary"] = {"error": repr(e)[:300]}
        if dog is None:
            print(json.dumps(res), flush=True)
        else:
            dog.emit(res)                            # (cancels the watchdog's own print; the shutdown below is still covered)
    if world > 1:
        from linna_amd import dist as ldist
        _at("shutdown")
        if rank == 0:
            # the line is out: a stuck teardown must not hold the launcher -- but it is a hang, and the exit code says so
            threading_guard = __import__("threading").Timer(float(os.environ.get("LINNA_BENCH_TEARDOWN_S", "120")),
                                                            lambda: os._exit(_Watchdog.EXIT_HANG))
            threading_guard.daemon = True
            threading_guard.start()
        ldist.shutdown()


_DOG = None

if __name__ == "__main__":
    try:
        main()
    except Exception as e:                                   # noqa: BLE001
        if _DOG is None:
            raise
        _DOG.failed(e)                                       # N > 1, behind the headline: the line is not lost to it
