#!/usr/bin/env python
"""Randomised checks of the one-call slice half step: over ensemble sizes, engines, schedules (one or several stepping-out
rounds, any trial counts) and networks, the chain under every linna_slice_fusion mask equals the chain under mask 0, and
both equal the round-by-round loop's as long as no walker overflows its rounds (then the guarded step has redone the
iteration on the round loop, and the chains must still agree).  usage: fuzz_slice.py [n] [seed0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np, torch
from test_gpu_serving import build_logprob
from linna_amd import sampler, _lib


def run(n, seed0):
    bad = 0
    lps = {name: build_logprob(name, 2.0)[0] for name in ("mlp_33_33", "v2_33_33")}
    prev = _lib.slice_fusion(-1)
    for it in range(n):
        rs = np.random.RandomState(seed0 + it)
        name = str(rs.choice(list(lps)))
        nw = 2 * int(rs.choice([3, 8, 17, 32, 33, 64, 65, 100, 128, 300, 512, 700]))
        rows = int(rs.choice([4, 8, 16]))
        nexp = int(rs.choice([1, 1, 1, 2, 3]))
        m_sched = [int(rs.randint(1, 17)) for _ in range(nexp)]
        nt_sched = [int(rs.randint(1, 33)) for _ in range(int(rs.randint(1, 4)))]
        mu = float(rs.choice([0.3, 0.7, 1.2]))
        x0 = (0.3 * rs.standard_normal((nw, 33))).astype(np.float32)
        tag = "cfg %d: %s nw %d rows %d m %s nt %s mu %.1f" % (seed0 + it, name, nw, rows, m_sched, nt_sched, mu)
        _lib.engine_rows(rows)
        try:
            outs = {}
            for mask in (0, 1, 2, 3):
                _lib.slice_fusion(mask)
                a = sampler.SliceEnsembleSampler(nw, 33, lps[name], seed=11, tune=False, mu=mu, fast=True)
                a.set_schedule(m_sched, nt_sched)
                a.set_state(x0)
                for _ in range(5):
                    a.step()
                torch.cuda.synchronize()
                outs[mask] = (a.coords.clone(), a.logp.clone(), a.noverflow)
            b = sampler.SliceEnsembleSampler(nw, 33, lps[name], seed=11, tune=False, mu=mu, fast=False)
            b.set_state(x0)
            for _ in range(5):
                b.step()
            torch.cuda.synchronize()
            ok = all(torch.equal(outs[m][0], outs[0][0]) and torch.equal(outs[m][1], outs[0][1]) for m in (1, 2, 3))
            ok = ok and torch.equal(outs[0][0], b.coords) and torch.equal(outs[0][1], b.logp)
            print("%s %s  overflows %s" % ("ok  " if ok else "BAD ", tag, [outs[m][2] for m in (0, 1, 2, 3)]), flush=True)
            bad += 0 if ok else 1
        finally:
            _lib.engine_rows(0)
            _lib.slice_fusion(prev)
    print("fuzz slice: %d configurations, %d bad" % (n, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
