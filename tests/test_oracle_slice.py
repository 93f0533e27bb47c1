"""CPU: the oracle's restatement of zeus' ensemble slice move (oracle/sampling.py ``slice_half_step``), the reference's
DEFAULT sampler (main.py:22 ``method="zeus"``, sampler.py:728-735).  zeus-mcmc is absent from the reference tree and from
this image and the reference holds no zeus chain: PARITY UNPINNED.  What can be checked without zeus is checked here --
the move's invariants, the published procedure's bookkeeping (Karamanis & Beutler 2021, Algorithms 2-4; Neal 2003, Fig. 3)
against a plain per-walker re-derivation written the way the paper states it, and the tuning rule."""
import numpy as np
import pytest

from oracle import sampling


def _gauss(nd, seed=0):
    rs = np.random.RandomState(seed)
    A = rs.standard_normal((nd, nd))
    cov = A @ A.T / nd + 0.3 * np.eye(nd)
    mean = rs.standard_normal(nd)
    ic = np.linalg.inv(cov)
    f = lambda q: (-0.5 * np.einsum("bi,ij,bj->b", q - mean, ic, q - mean)).astype(np.float32)
    return mean, cov, f


def test_slice_move_leaves_a_correlated_gaussian_invariant():
    """Started in zeus' tiny ball (util.py:937: 1e-3), tuned by the rule, the chain must carry the analytic mean and
    covariance; tuning must end with an expansion fraction near one half."""
    nd, nw = 5, 40
    mean, cov, f = _gauss(nd)
    rs = np.random.RandomState(1)
    x = (1e-3 * rs.standard_normal((nw, nd))).astype(np.float32)
    lp = f(x)
    mu, cnt, tune, tuned_at = 1.0, 0, True, None
    keep, fracs = [], []
    for it in range(2500):
        idx = rs.permutation(nw)
        x, lp, e, c = sampling.slice_iteration(x, lp, [idx[:nw // 2], idx[nw // 2:]], mu, 123, it, f)
        if tune:
            mu, cnt, tune = sampling.slice_tune_mu(mu, e, c, cnt)
            if not tune:
                tuned_at = it
        else:
            fracs.append(e / max(1, e + c))
        if it >= 400:
            keep.append(x.copy())
    assert tuned_at is not None and tuned_at < 300 and 0.1 < mu < 10
    assert abs(np.mean(fracs) - 0.5) < 0.08               # what the Robbins-Monro rule steers to
    s = np.concatenate(keep)
    assert np.all(np.abs(s.mean(0) - mean) / np.sqrt(np.diag(cov)) < 0.06)
    assert np.abs(np.cov(s.T) - cov).max() < 0.06 * np.abs(cov).max()
    np.testing.assert_allclose(lp, f(x), rtol=1e-6)        # the stored lnP is the lnP of the stored position


def _walker_by_the_paper(x, d, z0, l, J, K, us, f):
    """One walker, one slice update, written as Neal (2003) Fig. 3 + Fig. 5 state it (scalar loops, float64 bookkeeping of
    the SAME float32 quantities): returns (x', lnP(x'), L, R, nexp, ncon)."""
    f32 = np.float32
    r = f32(l + f32(1))
    nexp = ncon = 0
    while J >= 1 and z0 < f((x + l * d)[None].astype(f32))[0]:
        l = f32(l - f32(1)); J -= 1; nexp += 1
    while K >= 1 and z0 < f((x + r * d)[None].astype(f32))[0]:
        r = f32(r + f32(1)); K -= 1; nexp += 1
    for u in us:
        w = f32(l + u * f32(r - l))
        xp = (x + w * d).astype(f32)
        zp = f(xp[None])[0]
        if z0 < zp:
            return xp, zp, l, r, nexp, ncon
        if w < 0:
            l = w; ncon += 1
        elif w > 0:
            r = w; ncon += 1
    raise AssertionError("ran out of uniforms")


@pytest.mark.parametrize("maxsteps", [10000, 3])
def test_half_step_equals_the_per_walker_procedure_of_the_paper(maxsteps):
    """The batched half step (every walker's open bracket ends in one evaluation per pass, as zeus does) against the
    procedure applied walker by walker with the same draws: positions, lnP, brackets and both counts must be EQUAL -- also
    with a stepping-out budget of 3 (J + K = 2), which binds for most walkers at a small mu."""
    nd, nw = 4, 24
    mean, cov, f = _gauss(nd, 3)
    rs = np.random.RandomState(5)
    x = rs.multivariate_normal(mean, cov, nw).astype(np.float32)
    lp = f(x)
    S, C = np.arange(0, nw, 2), np.arange(1, nw, 2)
    seed, step, half, mu = 77, 9, 1, 0.08
    tr = {}
    x1, lp1, nexp, ncon = sampling.slice_half_step(x, lp, S, C, mu, seed, step, half, f, maxsteps=maxsteps, trace=tr)
    b0 = sampling.walker_bits(seed, S, step, half, 0)
    b1 = sampling.walker_bits(seed, S, step, half, 1)
    te = tc = 0
    for k, w in enumerate(S):
        ia = int((int(b0[k, 0]) * len(C)) >> 32)
        ib = int((int(b0[k, 1]) * (len(C) - 1)) >> 32)
        ib += ib >= ia
        assert ia != ib
        d = (np.float32(2 * mu) * (x[C[ia]] - x[C[ib]])).astype(np.float32)
        z0 = np.float32(lp[w] + np.log(sampling.u01(b0[k, 2])))
        l = np.float32(-sampling.u01(b0[k, 3]))
        J = min(int(np.floor(np.float32(maxsteps) * sampling.u01(b1[k, 0]))), maxsteps - 1)
        K = maxsteps - 1 - J
        us = [sampling.u01(sampling.walker_bits(seed, [w], step, 2 + half, t)[0, 0]) for t in range(1, 200)]
        xp, zp, L, R, e, c = _walker_by_the_paper(x[w], d, z0, l, J, K, us, f)
        assert np.array_equal(xp, x1[w]) and zp == lp1[w]
        assert L == tr["L"][k] and R == tr["R"][k] and e == tr["nexp"][k] and c == tr["ncon"][k]
        assert z0 < zp                                       # the accepted point lies in the slice
        assert tr["L"][k] < tr["W"][k] < tr["R"][k] and tr["L"][k] < 0 < tr["R"][k]   # ... inside a bracket around the walker
        te += e; tc += c
    assert (te, tc) == (nexp, ncon)
    assert np.array_equal(x1[C], x[C]) and np.array_equal(lp1[C], lp[C])   # the complementary half does not move
    if maxsteps == 3:
        assert tr["nexp"].max() <= 2 and (tr["nexp"] == 2).sum() >= 4      # the budget binds, and never more than J + K steps
    else:
        assert tr["nexp"].max() > 2


def test_slice_heights_brackets_and_budgets_have_their_laws():
    """Z0 - lnP = log u is -Exp(1); L ~ -U(0,1), R = L + 1; J ~ floor(maxsteps U) in [0, maxsteps - 1], J + K = maxsteps - 1
    (zeus ensemble.py; Neal 2003: the budget m split uniformly)."""
    n = 20000
    S = np.arange(n)
    b0 = sampling.walker_bits(5, S, 1, 0, 0)
    e = -np.log(sampling.u01(b0[:, 2]).astype(np.float64))
    assert abs(e.mean() - 1) < 0.03 and abs(e.var() - 1) < 0.08
    l = -sampling.u01(b0[:, 3])
    assert -1 < l.min() and l.max() < 0 and abs(l.mean() + 0.5) < 0.01
    tr = {}
    f = lambda q: np.zeros(len(q), np.float32) - np.float32(1e30) * (np.abs(q).max(-1) > 1e-3)
    x = np.zeros((n + 2, 1), np.float32); x[n] = 1e-6; x[n + 1] = -1e-6
    sampling.slice_half_step(x, f(x), S, np.array([n, n + 1]), 1.0, 5, 1, 0, f, maxsteps=7, trace=tr)
    J0 = np.minimum(np.floor(np.float32(7) * sampling.u01(sampling.walker_bits(5, S, 1, 0, 1)[:, 0])), 6)
    assert J0.min() == 0 and J0.max() == 6 and np.abs(np.bincount(J0.astype(int)) / n - 1 / 7).max() < 0.01
    # a flat density under the slice everywhere near the walker: stepping out always uses the whole budget, J left and K right
    assert np.array_equal(tr["nexp"], np.full(n, 6)) and np.all(tr["J"] == 0) and np.all(tr["K"] == 0)
    np.testing.assert_array_equal(np.round(tr["R_out"] - tr["L_out"]), 7)


def test_tuning_rule_and_limits():
    mu, cnt, tune = sampling.slice_tune_mu(1.0, 0, 10, 0)                  # nexp = max(1, nexp)
    assert mu == pytest.approx(2.0 / 11) and cnt == 0 and tune
    mu, cnt, tune = 1.0, 0, True
    for i in range(6):
        mu, cnt, tune = sampling.slice_tune_mu(mu, 50, 52, cnt)           # |50/102 - 1/2| < 0.05: six in a row end tuning
        assert tune == (i < 5)
    assert mu == pytest.approx((100 / 102) ** 6)
    assert sampling.slice_tune_mu(1.0, 10, 30, 5)[1] == 0                  # a miss resets the count
    # zeus raises when a walker needs more than `maxiter` passes: a density that is -inf everywhere can never accept
    f = lambda q: np.full(len(q), -np.inf, np.float32)
    x = np.random.RandomState(0).standard_normal((6, 2)).astype(np.float32)
    with pytest.raises(sampling.SliceLimit):
        sampling.slice_half_step(x, np.zeros(6, np.float32), np.arange(3), np.arange(3, 6), 1.0, 1, 0, 0, f, maxiter=50)
