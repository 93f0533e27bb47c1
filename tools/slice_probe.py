"""Timing target: zeus-style ensemble slice sampler iterations on the bench problem, round-by-round loop against the
one-call half step (linna_slice_half_step), over ensemble sizes.  usage: slice_probe.py [nw ...]
SLICE_FIRST=a,b,...: the one-call path with that many bracket ends per side in its first round (SliceEnsembleSampler.FAST_FIRST; 0: by ensemble size)
SLICE_ONLY_FAST=1: skip the round loop
SLICE_SCHED="8/16,16;8/16;8/32": the one-call path under each of these schedules (ends per side by round / trials by round)
SLICE_FUSION=0,1,7: under each of these linna_slice_fusion masks"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from linna_amd import sampler, _lib
sizes = [int(a) for a in sys.argv[1:]] or [16, 128, 512, 1024, 4096]
points = [int(a) for a in os.environ.get("SLICE_FIRST", "0").split(",")]
lp, model, consts = bench.build_problem(torch.device("cuda", 0))
scheds = [tuple([int(v) for v in part.split(",")] for part in s.split("/")) for s in os.environ.get("SLICE_SCHED", "").split(";") if s]
masks = [int(v) for v in os.environ.get("SLICE_FUSION", "").split(",") if v] or [None]
for nw in sizes:
    for fast, pts, mask in ([] if os.environ.get("SLICE_ONLY_FAST") else [(False, 0, None)]) + [(True, p, f) for p in (scheds or points) for f in masks]:
        if mask is not None:
            _lib.slice_fusion(mask & 7)
            sampler.SliceEnsembleSampler.USE_EXPECT = not (mask & 8)      # (bit 3 of SLICE_FUSION: engines of the later rounds by the fixed rule)
        sampler.SliceEnsembleSampler.FAST_FIRST = (pts or None) if not scheds else None
        ens = sampler.SliceEnsembleSampler(nw, 33, lp, seed=1, fast=fast)
        if fast and scheds:
            ens.set_schedule(*pts)
        ens.set_state(0.05 * np.random.RandomState(7).standard_normal((nw, 33)))
        ens.run(60, store=False)
        torch.cuda.synchronize()
        n = 400 if nw <= 1024 else 60
        e0 = ens.neval
        t0 = time.perf_counter()
        ens.run(n, store=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%5d walkers  %-44s %8.1f us/iteration  %7.0f it/s  mu %.3f  evals/walker/iteration %.1f  tuned %s" % (
            nw, ("one-call m %s nt %s%s" % (ens.m_sched, ens.nt_sched, "" if mask is None else " fusion %d" % mask)) if fast else "rounds", dt / n * 1e6, n / dt, ens.mu, (ens.neval - e0) / n / nw, not ens.tune), flush=True)
        if fast:
            print("        expected trial points by round (engine choice of the later rounds): %s after %d one-call iterations" % (ens.expected_rows, getattr(ens, "_fast_steps", 0)), flush=True)
        if fast and ens.round_usage():
            u = ens.round_usage()
            print("        still active behind each stepping-out round: %s; behind each shrinking round: %s (mean fraction of a half ensemble, %d half steps); runs redone on the round loop: %d" % (
                ["%.2e" % v for v in u["active_after_expand_round"]], ["%.2e" % v for v in u["active_after_shrink_round"]], u["half_steps"], ens.noverflow), flush=True)
