"""GPU: the ensemble slice sampler's kernels (the reference's DEFAULT sampler: zeus behind sampler.py:728-735, main.py:22)
replayed half step by half step against the oracle's restatement of zeus' move (oracle/sampling.py ``slice_half_step``).

Both sides make the SAME Philox draws and walk the same procedure; they differ in who evaluates lnP (the HIP whole-network
kernel in fp32 against the numpy oracle emulator).  The replay is in LOCKSTEP: before every half step the oracle takes the
ensemble's positions and lnP from the device, both sides advance one half step, and every walker is compared -- slice
height, the bracket [L, R] the shrinking ended with, the accepted weight, the new position and its lnP, and the half
ensemble's expansion / contraction counts (which tune mu).  A discrete decision ``Z0 < lnP(x)`` can only differ where
|lnP - Z0| is at the rounding level of lnP: a walker is EXEMPT from the exact comparison iff the oracle saw a comparison of
its with a margin below the serving goldens' tolerance (2e-5 |lnP|, tests/test_gpu_serving.py) -- a fraction of a per cent of
the walker half steps, asserted below -- and every other walker must agree EXACTLY in its discrete outcome.
zeus itself is absent (third-party, unpinned, no fixture in the reference): parity of the oracle is unpinned, see its header."""
import numpy as np
import pytest

import cases
from linna_amd import _lib

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
from test_gpu_serving import build_logprob  # noqa: E402

LNP_RTOL = 2e-5                      # lnP of the HIP path against the oracle's: the serving goldens' tolerance
GOLDEN = 0x9E3779B97F4A7C15


class Replay(object):
    """Drives one SliceEnsembleSampler in lockstep with the oracle and keeps the tallies."""

    def __init__(self, ens, prob, temperature, seed):
        from oracle import likelihood
        emu = cases.oracle_emulator(prob)
        self.f = lambda q: likelihood.log_prob(q, emu, prob["priors"], prob["data"], prob["invcov"], temperature)
        self.ens, self.nd = ens, ens.ndim
        self.lib_seed = (seed + GOLDEN * 1) & 0xFFFFFFFFFFFFFFFF
        self.before = None
        self.walker_half_steps = self.exempt = self.unfinished = 0
        self.worst_lnp = 0.0
        self.mu_checked = self.mu_exact = 0
        self.count_mismatch_budget = 0
        self.gpu_counts = np.zeros(2, np.int64)
        self.ora_counts = np.zeros(2, np.int64)
        self.paths = set()
        self.half_log = []
        self.max_expansions = 0
        self.budget_bound = 0
        ens.probe = self.after_half
        self._orig_splits = ens._splits
        ens._splits = self._splits

    def _splits(self):
        h = self._orig_splits()
        self.halves = h.cpu().numpy().astype(np.int64)
        self.snapshot()
        return h

    def snapshot(self):
        torch.cuda.synchronize()
        e = self.ens
        self.before = (e.coords[:, :self.nd].cpu().numpy().copy(), e.logp.cpu().numpy().copy(), int(e.step_dev.item()), e.mu)
        self.prev_counters = None

    def after_half(self, h, S, st):
        from oracle import sampling
        torch.cuda.synchronize()
        e = self.ens
        x0, lp0, step, mu = self.before
        Sx, Cx = self.halves[h], self.halves[1 - h]
        assert np.array_equal(S.cpu().numpy(), Sx)
        tr = {}
        x1, lp1, nexp, ncon = sampling.slice_half_step(x0, lp0, Sx, Cx, mu, self.lib_seed, step, h, self.f, maxsteps=e.maxsteps, trace=tr)
        g = {k: st[k].cpu().numpy().copy() for k in ("Z0", "L", "R", "Wacc", "Zacc")}
        xg, lg = e.coords[:, :self.nd].cpu().numpy(), e.logp.cpu().numpy()
        self.paths.add("one-call" if st["fast"] else "rounds")
        # slice heights: lnP + log u of the same uniform (logf on the device against numpy's: an ulp or two)
        np.testing.assert_allclose(g["Z0"], tr["Z0"], rtol=1e-6, atol=1e-6)
        tol = LNP_RTOL * np.maximum(1.0, np.abs(tr["Z0"].astype(np.float64))) + 2e-6
        clean = tr["margin"] > tol
        ns = len(Sx)
        if st["fast"]:
            # a walker that needed more stepping-out steps / trials than the call's speculative rounds hold is left in place and
            # counted (counters[2]); the product then redoes the run on the round loop (SliceEnsembleSampler._guard) -- here it
            # is set aside and counted
            unfinished = e.flags[:3 * ns].cpu().numpy().reshape(ns, 3).any(axis=1)
            self.unfinished += int(unfinished.sum())
            clean &= ~unfinished
        self.walker_half_steps += ns
        self.exempt += int((~clean).sum())
        # every clean walker: the same discrete outcome, exactly
        for key in ("L", "R"):
            assert np.array_equal(g[key][clean], tr[key][clean]), (key, h, np.flatnonzero(g[key] != tr[key]), clean.sum())
        assert np.array_equal(g["Wacc"][clean], tr["W"][clean])
        assert np.all(tr["L"][clean] < tr["W"][clean]) and np.all(tr["W"][clean] < tr["R"][clean])
        rel = np.abs(g["Zacc"][clean] - tr["Zacc"][clean]) / np.maximum(1.0, np.abs(tr["Zacc"][clean]))
        self.worst_lnp = max(self.worst_lnp, float(rel.max()) if clean.any() else 0.0)
        assert np.all(rel <= LNP_RTOL), rel.max()
        np.testing.assert_allclose(xg[Sx[clean]], x1[Sx[clean]], rtol=3e-7, atol=1e-7)     # x + W d, the same float32 operations
        np.testing.assert_allclose(lg[Sx[clean]], lp1[Sx[clean]], rtol=LNP_RTOL)
        assert np.array_equal(xg[Cx], x0[Cx]) and np.array_equal(lg[Cx], lp0[Cx])          # the complementary half stands still
        # an accepted point lies in its slice on the DEVICE's own numbers too
        assert np.all(g["Z0"][clean] < g["Zacc"][clean])
        # expansion / contraction counts of this half step (the device counters run over the iteration's two half steps)
        c = st["counters"][:2].cpu().numpy().astype(np.int64)
        if h == 1 and self.prev_counters is not None:
            c = c - self.prev_counters
        else:
            self.prev_counters = c.copy()
        if clean.all():
            assert (c[0], c[1]) == (nexp, ncon), (h, c, nexp, ncon)
        else:
            slack = int((tr["nexp"][~clean] + tr["ncon"][~clean]).sum()) + 40 * int((~clean).sum())
            assert abs(int(c[0]) - nexp) + abs(int(c[1]) - ncon) <= slack, (c, nexp, ncon, slack)
        self.gpu_counts += c
        self.ora_counts += (nexp, ncon)
        self.half_log.append((nexp, ncon, bool(clean.all())))
        self.max_expansions = max(self.max_expansions, int(tr["nexp"].max()))
        self.budget_bound += int(((tr["J"] == 0) | (tr["K"] == 0)).sum())
        # lockstep: the next half step starts from the device's state
        self.before = (xg.copy(), lg.copy(), step, mu)


def _run(name, T, nw, seed, iters, x_scale, prepare, **kw):
    from linna_amd import sampler
    lp, pred, yinv, prob = build_logprob(name, T)
    nd = prob["nin"]
    x0 = (x_scale * np.random.RandomState(nw + 7).standard_normal((nw, nd))).astype(np.float32)
    ens = sampler.SliceEnsembleSampler(nw, nd, lp, seed=seed, **kw)
    prepare(ens)
    ens.set_state(x0)
    rp = Replay(ens, prob, T, seed)
    from oracle import sampling
    mus = []
    for it in range(iters):
        mu0, cnt0, tuning = ens.mu, ens._tune_count, ens.tune
        ens._step()                                  # (no overflow guard: a redone run would replay its iterations twice)
        mus.append(ens.mu)
        if tuning:
            # zeus' rule on the ORACLE's counts of this iteration, from the mu both sides started it with
            (e0, c0, ok0), (e1, c1, ok1) = rp.half_log[-2:]
            mu1, cnt1, still = sampling.slice_tune_mu(mu0, e0 + e1, c0 + c1, cnt0)
            if ok0 and ok1:
                assert ens.mu == mu1 and ens._tune_count == cnt1 and ens.tune == still, (it, ens.mu, mu1)
            else:
                assert abs(ens.mu - mu1) <= 0.05 * mu1, (it, ens.mu, mu1)
            rp.mu_checked += 1
            rp.mu_exact += int(ok0 and ok1)
        else:
            assert ens.mu == mu0
    torch.cuda.synchronize()
    return ens, rp, mus


@pytest.mark.parametrize("name,T,nw", [("mlp_33_33", 2.0, 34), ("mlp_33_33", 2.0, 128), ("mlp_33_33", 2.0, 1024),
                                       ("v2_33_33", 2.0, 34), ("v2_33_33", 2.0, 128), ("v2_33_33", 2.0, 1024),
                                       ("mlp_7_5_small", 32.0, 34), ("mlp_7_5_small", 32.0, 128), ("mlp_7_5_small", 32.0, 1024)])
def test_slice_half_steps_replay_against_the_oracle(name, T, nw):
    """20 iterations at a fixed mu through the round-by-round entries and through linna_slice_half_step under every
    linna_slice_fusion mask (with ONE stepping-out round, so that every fold of the masks is live, and with the ensemble's own
    multi-round schedule)."""
    prev = _lib.slice_fusion(-1)
    one_round = lambda e: e.set_schedule([8], [16, 16])
    paths = [("rounds", dict(fast=False), lambda e: None, None)]
    paths += [("one-call, one stepping-out round, fusion %d" % m, dict(fast=True), one_round, m) for m in (0, 1, 3, 7)]
    paths += [("one-call, default schedule", dict(fast=True), lambda e: None, prev)]
    try:
        for label, kw, prepare, mask in paths:
            if mask is not None:
                _lib.slice_fusion(mask)
            ens, rp, _ = _run(name, T, nw, seed=31, iters=20, x_scale=0.3, prepare=prepare, tune=False, mu=0.45, **kw)
            want = "rounds" if kw["fast"] is False else "one-call"
            assert rp.paths == {want}, (label, rp.paths)
            assert rp.walker_half_steps == 20 * nw
            assert rp.exempt <= 0.02 * rp.walker_half_steps + 2, (label, rp.exempt, rp.walker_half_steps)
            assert rp.gpu_counts[0] > 0 and rp.gpu_counts[1] > 0
            if kw["fast"]:
                assert int(ens._fast_bufs["counters"][2].item()) == rp.unfinished, label
                assert rp.unfinished <= 0.002 * rp.walker_half_steps, (label, rp.unfinished)     # (the rounds of a call almost always suffice)
            print("%-48s %-14s nw %4d: exempt %d of %d walker half steps, worst |dlnP|/|lnP| %.1e, expansions %d / contractions %d "
                  "(oracle %d / %d)%s" % (label, name, nw, rp.exempt, rp.walker_half_steps, rp.worst_lnp, rp.gpu_counts[0],
                                          rp.gpu_counts[1], rp.ora_counts[0], rp.ora_counts[1],
                                          ", %d left unfinished by the call's rounds" % rp.unfinished if rp.unfinished else ""))
    finally:
        _lib.slice_fusion(prev)


@pytest.mark.parametrize("name,T,nw", [("mlp_33_33", 2.0, 34), ("v2_33_33", 2.0, 128), ("mlp_7_5_small", 32.0, 128)])
def test_tuning_from_the_tiny_ball_replays_with_the_same_mu(name, T, nw):
    """The start of every run (util.py:937: walkers in a 1e-3 ball around the initial point, mu = 1): hundreds of stepping-out
    steps per side in the first iterations (the round loop), mu tuned by zeus' rule from the ensemble's counts, then the
    product's own switch to the one-call path -- 40 iterations in lockstep; mu must follow the oracle's rule on the oracle's
    counts: exactly while no walker was exempt, and within what the exempt walkers' counts can move it afterwards."""
    from oracle import sampling
    ens, rp, mus = _run(name, T, nw, seed=5, iters=40, x_scale=1e-3, prepare=lambda e: None, tune=True)
    assert rp.paths == {"rounds", "one-call"}, rp.paths
    assert rp.max_expansions > 50                         # the tiny ball really was stepped out of
    assert rp.exempt <= 0.02 * rp.walker_half_steps + 2
    assert abs(int(rp.gpu_counts[0]) - int(rp.ora_counts[0])) <= 0.01 * rp.ora_counts[0] + 80 * rp.exempt
    assert 0.02 < mus[-1] < 20
    assert rp.mu_checked >= 10 and rp.mu_exact >= 3, (rp.mu_checked, rp.mu_exact)    # (exact wherever no walker of the iteration was exempt)
    print("%s nw %d: mu %.4f after 40 iterations (tuning %s), most expansions of one walker %d, walkers whose budget J or K "
          "ran out %d, exempt %d of %d; mu equal to the oracle's rule on the oracle's counts in %d of %d tuning iterations, within 5 %% in the rest"
          % (name, nw, mus[-1], "on" if ens.tune else "off at %d" % ens.tune_off_iteration, rp.max_expansions, rp.budget_bound,
             rp.exempt, rp.walker_half_steps, rp.mu_exact, rp.mu_checked))


def test_the_stepping_out_budget_binds_on_the_device_as_in_the_oracle():
    """maxsteps = 4 (J + K = 3) at a small mu: most walkers spend their whole budget; the kernels (round loop and one-call
    path with its speculative rounds of 8 ends per side) must stop where zeus stops -- brackets equal, walker by walker."""
    for kw, prepare in [(dict(fast=False), lambda e: None), (dict(fast=True), lambda e: e.set_schedule([8], [16, 16])),
                        (dict(fast=True), lambda e: e.set_schedule([1, 2, 4], [4, 8, 16]))]:
        ens, rp, _ = _run("mlp_33_33", 2.0, 128, seed=3, iters=6, x_scale=0.3, prepare=prepare, tune=False, mu=0.02, maxsteps=4, **kw)
        assert rp.max_expansions <= 3 and rp.budget_bound > 0.5 * rp.walker_half_steps, (rp.max_expansions, rp.budget_bound)
        assert rp.exempt <= 0.03 * rp.walker_half_steps + 2
