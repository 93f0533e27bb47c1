"""GPU: the incremental convergence statistics (csrc/autocorr.hip through the C ABI: running lagged products ->
emcee's integrated autocorrelation time; checkmeanstd's moments) against the oracle's restatement of emcee's estimator
(oracle/sampling.py: FFT over the whole chain) on the reference-held emcee chain and on synthetic AR(1) chains, and the
block entry of the stretch move (linna_stretch_run) against the per-iteration loop.

Tolerance: the two routes compute the same sums in float64 in a different order (direct lagged products with the mean
taken out through prefix sums vs a zero-padded FFT): relative differences of the autocovariances are a few 1e-13; the
asserted 1e-9 is the VERDICT's bar.  The window index is discrete and must agree exactly."""
import os

import numpy as np
import pytest

import cases
from linna_amd import _lib

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

FIXTURE = os.path.join(cases.GOLDEN, "2dgaussian_Fulltconn", "iter_0", "chemcee_256.h5")


def ar1(nt, nw, rho, seed, mean=0.0, scale=1.0):
    rs = np.random.RandomState(seed)
    rho = np.asarray(rho, np.float64)
    x = np.zeros((nt, nw, len(rho)))
    e = rs.standard_normal((nt, nw, len(rho)))
    x[0] = e[0] / np.sqrt(1 - rho ** 2)
    for i in range(1, nt):
        x[i] = rho * x[i - 1] + e[i]
    return (mean + scale * x).astype(np.float32)


def test_tau_on_the_reference_held_emcee_chain():
    """chemcee_256.h5 (emcee 3.0.2 through h5py, 200 steps x 4 walkers x 2 parameters): the running-sum estimate equals
    the oracle's, whole chain and block by block as a run sees it."""
    from oracle import sampling
    from linna_amd import sampler
    d = sampler.ChainStore.read_h5(FIXTURE)
    z = np.asarray(d["chain"], np.float32)
    assert z.shape == (200, 4, 2)
    ref = sampling.integrated_time(z.astype(np.float64))
    dc = sampler.DeviceChain()
    dc.append(torch.as_tensor(z, device="cuda"))
    np.testing.assert_allclose(dc.integrated_time(), ref, rtol=1e-9)
    np.testing.assert_allclose(dc.integrated_time(), sampler.integrated_time(z.astype(np.float64)), rtol=1e-9)
    dc2 = sampler.DeviceChain()
    for n0 in range(0, 200, 50):
        dc2.append(torch.as_tensor(z[n0:n0 + 50], device="cuda"))
        np.testing.assert_allclose(dc2.integrated_time(), sampling.integrated_time(z[:n0 + 50].astype(np.float64)), rtol=1e-9)
    np.testing.assert_allclose(dc2.integrated_time(discard=40), sampling.integrated_time(z[40:].astype(np.float64)), rtol=1e-9)


@pytest.mark.parametrize("nw", [24, 64, 100, 130])
def test_incremental_tau_matches_the_oracle_on_ar1_chains(nw):
    """Blocks of 100 as in the driver loop; walkers not a multiple of the 64-lane padding; a large common offset and a small
    scale (the cancellation the per-series reference point removes); every check equals the oracle on the chain so far."""
    from oracle import sampling
    from linna_amd import sampler
    rho = np.array([0.5, 0.9, 0.97])
    x = ar1(1200, nw, rho, seed=nw, mean=np.array([0.0, 30.0, -7.0]), scale=np.array([1.0, 1e-2, 3.0]))
    dc = sampler.DeviceChain()
    for n0 in range(0, 1200, 100):
        dc.append(torch.as_tensor(x[n0:n0 + 100], device="cuda"))
        tau = dc.integrated_time()
        if n0 in (0, 100, 500, 1100):
            np.testing.assert_allclose(tau, sampling.integrated_time(x[:n0 + 100].astype(np.float64)), rtol=1e-9)
    assert np.all(np.abs(tau / ((1 + rho) / (1 - rho)) - 1) < 0.5)                # AR(1): tau = (1 + rho) / (1 - rho)
    xf = x.astype(np.float64)
    np.testing.assert_allclose(dc.integrated_time(upto=450), sampling.integrated_time(xf[:450]), rtol=1e-9)
    np.testing.assert_allclose(dc.integrated_time(discard=90, upto=450), sampling.integrated_time(xf[90:450]), rtol=1e-9)
    np.testing.assert_allclose(dc.integrated_time(), sampling.integrated_time(xf), rtol=1e-9)
    assert dc.last(250).shape == (250, nw, 3) and torch.equal(dc.last(250).cpu(), torch.as_tensor(x[-250:]))


def test_moving_discard_and_lag_growth():
    """zeus' callback drops the first 20 % of the chain at every check (sampler.py:684): rows leave the running sums at
    the front while others enter at the end.  A strongly correlated chain needs more lags than the initial 512: they are
    computed from the stored chain on demand and the estimate still equals the oracle's."""
    from oracle import sampling
    from linna_amd import sampler
    x = ar1(6000, 16, [0.6, 0.995], seed=3)
    dc = sampler.DeviceChain()
    for n0 in range(0, 6000, 100):
        dc.append(torch.as_tensor(x[n0:n0 + 100], device="cuda"))
        done = n0 + 100
        tau = dc.integrated_time(discard=int(done * 0.2))
        if done in (100, 1000, 3000, 6000):
            np.testing.assert_allclose(tau, sampling.integrated_time(x[int(done * 0.2):done].astype(np.float64)), rtol=1e-9)
    assert dc.lag_growths >= 1 and len(dc._S) > 512
    assert tau[1] > 150


def test_nan_for_a_walker_that_never_moved_and_short_chains():
    """emcee's estimator returns NaN when a series is constant (0 / 0 in the normalisation): the driver's NaN rule
    (sampler.py:542-543) depends on it.  Chains of 1..3 steps work as the oracle does."""
    from oracle import sampling
    from linna_amd import sampler
    x = ar1(300, 8, [0.5, 0.7], seed=5)
    x[:, 3, 1] = x[0, 3, 1]
    dc = sampler.DeviceChain()
    dc.append(torch.as_tensor(x, device="cuda"))
    tau = dc.integrated_time()
    with np.errstate(invalid="ignore"):
        ref = sampling.integrated_time(x.astype(np.float64))
    assert np.isnan(tau[1]) and np.isnan(ref[1])
    np.testing.assert_allclose(tau[0], ref[0], rtol=1e-9)
    for n in (2, 3, 33):
        dc = sampler.DeviceChain()
        dc.append(torch.as_tensor(x[:n, :, :1], device="cuda"))
        np.testing.assert_allclose(dc.integrated_time(), sampling.integrated_time(x[:n, :, :1].astype(np.float64)), rtol=1e-9, atol=1e-12)


def test_walker_subset_and_checkmeanstd():
    from oracle import sampling
    from linna_amd import sampler
    x = ar1(700, 96, [0.5, 0.8, 0.9], seed=9, mean=np.array([1.0, -2.0, 0.5]))
    x[350:] += np.float32(0.05)                               # a drift the mean / std comparison sees
    dc = sampler.DeviceChain(max_walkers=32)                  # every third walker
    dc.append(torch.as_tensor(x[:300], device="cuda"))
    dc.append(torch.as_tensor(x[300:], device="cuda"))
    assert dc.wstride == 3 and dc.nws == 32 and dc.subset
    xf = x.astype(np.float64)
    np.testing.assert_allclose(dc.integrated_time(), sampling.integrated_time(xf[:, ::3]), rtol=1e-9)
    # the drivers' confirmation: every walker of the ensemble, sums of their own from the stored chain
    np.testing.assert_allclose(dc.integrated_time(all_walkers=True), sampling.integrated_time(xf), rtol=1e-9)
    np.testing.assert_allclose(dc.integrated_time(upto=600, all_walkers=True), sampling.integrated_time(xf[:600]), rtol=1e-9)
    np.testing.assert_allclose(dc.integrated_time(discard=140, all_walkers=True), sampling.integrated_time(xf[140:]), rtol=1e-9)
    np.testing.assert_allclose(dc.integrated_time(), sampling.integrated_time(xf[:, ::3]), rtol=1e-9)     # the running sums are untouched
    assert torch.equal(dc.last(77).cpu(), torch.as_tensor(x[-77:]))                                      # lanes back in walker order
    d2 = sampler.DeviceChain(max_walkers=50)                  # 130 walkers -> every third: 44 of them, 86 others behind
    y = ar1(400, 130, [0.7], seed=1)
    d2.append(torch.as_tensor(y, device="cuda"))
    assert d2.wstride == 3 and d2.nws == 44 and d2.nwc == 64 and d2.nwp == 192
    np.testing.assert_allclose(d2.integrated_time(), sampling.integrated_time(y[:, ::3].astype(np.float64)), rtol=1e-9)
    np.testing.assert_allclose(d2.integrated_time(all_walkers=True), sampling.integrated_time(y.astype(np.float64)), rtol=1e-9)
    assert torch.equal(d2.last(5).cpu(), torch.as_tensor(y[-5:]))
    full = sampler.DeviceChain()
    full.append(torch.as_tensor(x, device="cuda"))
    import io, contextlib
    for n in (40, 333, 700):
        a, b = sampling.checkmeanstd_stats(x[-n:].astype(np.float64))
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            got = full.checkmeanstd(n, 0.03, 0.03)
        ga, gb = (float(v) for v in buf.getvalue().split())
        np.testing.assert_allclose([ga, gb], [a, b], rtol=1e-9, atol=1e-12)
        assert got == bool((a < 0.03) and (b < 0.03))
        with contextlib.redirect_stdout(io.StringIO()):
            assert got == bool(sampler.checkmeanstd(x[-n:].astype(np.float64), 0.03, 0.03))


def test_kernel_entries_reject_bad_shapes():
    from linna_amd import sampler
    lib = _lib.load()
    ctx = _lib.ctx()
    ct = torch.zeros((8, 2, 64), dtype=torch.float32, device="cuda")
    S = torch.zeros((32, 2, 64), dtype=torch.float64, device="cuda")
    T = torch.zeros((2, 64), dtype=torch.float64, device="cuda")
    P = lambda t: _lib.ptr(t, t.dtype)
    st = _lib.stream()
    assert lib.linna_acorr_update(ctx, P(ct), 2, 63, 63, 0, 8, 0, 8, 0, 32, P(S), P(T), 0, st) == -1    # walkers not padded to 64
    assert lib.linna_acorr_update(ctx, P(ct), 2, 64, 128, 0, 8, 0, 8, 0, 32, P(S), P(T), 0, st) == -1   # more covered lanes than the chain has
    assert lib.linna_acorr_update(ctx, P(ct), 2, 64, 64, 0, 8, 0, 8, 0, 30, P(S), P(T), 0, st) == -1    # lag range not a multiple of 32
    assert lib.linna_acorr_update(ctx, P(ct), 2, 64, 64, 0, 9, 0, 8, 0, 32, P(S), P(T), 0, st) == -1    # anchors outside the window
    out = torch.zeros(6, dtype=torch.float64, device="cuda")
    scr = torch.zeros(4096, dtype=torch.float64, device="cuda")
    assert lib.linna_acorr_tau(ctx, P(ct), 2, 64, 64, 4, 0, 8, 8, P(S), P(T), 5.0, P(scr), P(out), st) == -1   # kuse > N - 1
    assert lib.linna_acorr_tau(ctx, P(ct), 2, 64, 64, 65, 0, 8, 7, P(S), P(T), 5.0, P(scr), P(out), st) == -1  # more walkers than lanes
    assert lib.linna_acorr_update(ctx, P(ct), 2, 64, 64, 0, 8, 0, 8, 0, 32, P(S), P(T), 0, st) == 0
    torch.cuda.synchronize()


def _bench_lp():
    from test_gpu_serving import build_logprob
    return build_logprob("mlp_33_33", 1.0)


@pytest.mark.parametrize("nw", [8, 128, 600])
def test_block_run_is_bit_identical_to_the_iteration_loop(nw):
    """linna_stretch_run (one C call per block, chain rows written by the kernels' finish) against the host loop over
    linna_stretch_half_step with device copies of the state after every iteration: same chain, log-probabilities,
    acceptance counts, final state -- bit for bit; also when the two routes alternate on one sampler."""
    from linna_amd import sampler
    lp = _bench_lp()[0]
    x0 = 0.3 * np.random.RandomState(nw).standard_normal((nw, 33)).astype(np.float32)
    a = sampler.EnsembleSampler(nw, 33, lp, seed=5)
    b = sampler.EnsembleSampler(nw, 33, lp, seed=5)
    b.block_run = False
    a.set_state(x0); b.set_state(x0)
    ca, la = a.run(37)
    cb, lb = b.run(37)
    assert a.block_run is True and b.block_run is False
    assert torch.equal(ca, cb) and torch.equal(la, lb)
    a.step(); b.step()                                             # the per-iteration route continues the same split sequence
    ca, la = a.run(70); cb, lb = b.run(70)
    assert torch.equal(ca, cb) and torch.equal(la, lb)
    a.run(5, store=False); b.run(5, store=False)
    assert torch.equal(a.coords, b.coords) and torch.equal(a.logp, b.logp) and torch.equal(a.naccept, b.naccept)
    assert a.iteration == b.iteration == 113
    assert torch.equal(ca[-1], cb[-1]) and float(a.naccept.float().mean()) > 0
    c = sampler.EnsembleSampler(nw, 33, lp, seed=5, randomize_split=False)
    d = sampler.EnsembleSampler(nw, 33, lp, seed=5, randomize_split=False)
    d.block_run = False
    c.set_state(x0); d.set_state(x0)
    cc, lc = c.run(20); cd, ld = d.run(20)
    assert torch.equal(cc, cd) and torch.equal(lc, ld)


def test_pipelined_driver_stops_where_the_sequential_criterion_stops(tmp_path):
    """The emcee driver overlaps the statistics of block i with the sampling of block i + 1 and drops that block when the
    verdict is "stop": the stored chain must end at the first check that meets the reference's criterion
    (sampler.py:545-552) -- recomputed here with the oracle's estimator on the stored chain -- and no earlier check may."""
    import contextlib, io
    from oracle import sampling
    from linna_amd import sampler, util
    from test_gpu_sampling import identity_emulator_logprob
    nd, nw = 4, 32
    rs = np.random.RandomState(2)
    means, cov = rs.uniform(-0.2, 0.2, nd), np.diag(0.1 * rs.uniform(0.5, 1.0, nd))
    priors = [{"param": "p%d" % i, "dist": "flat", "arg1": -5.0, "arg2": 5.0} for i in range(nd)]
    lp = identity_emulator_logprob(nd, means, cov, priors)
    x0 = 1e-3 * rs.standard_normal((nw, nd))
    drv = sampler.HMCSampler(lp, None, None, nd, nw, x0=x0, transform=util.Transform(priors), seed=3)
    prof = {}
    with contextlib.redirect_stdout(io.StringIO()):
        store = drv.sample(None, 20000, outdir=str(tmp_path), ntimes=20, tautol=0.05, meanshift=0.2, stdshift=0.2, nk=2, profile=prof)
    z, th, l = store.arrays()
    n = len(z)
    assert 300 <= n < 20000 and n % 100 == 0 and prof["iterations"] == n
    zf = z.astype(np.float64)
    old = np.inf
    for done in range(100, n + 1, 100):
        tau = sampling.integrated_time(zf[:done])
        ok = np.all(tau * 20 < done) and np.all(np.abs(old - tau) / tau < 0.05)
        if ok:
            a, b = sampling.checkmeanstd_stats(zf[done - max(2, int(2 * np.mean(tau))):done])
            ok = (a < 0.2) and (b < 0.2)
        assert bool(ok) == (done == n), (done, n, tau)
        old = tau
    d = sampler.ChainStore.read_h5(os.path.join(str(tmp_path), "chemcee_256.h5"))
    assert d["iteration"] == n and np.array_equal(d["chain"], z)
