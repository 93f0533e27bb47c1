"""GPU: the callback surface of ``ml_sampler_core`` and the steps either side of training that
take user input: ``loglikelihoodfunc`` (main.py:277-279, util.py:953), ``externalloglike`` (util.py:1003-1008), a
``pool`` object (main.py:186, 282-286; util.py:258-289), gauss priors (main.py:125-126, util.py:339-343),
``params["nimp"]`` (main.py:297-334) and ``chisqcut`` (util.py:1260-1270)."""
import os

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
from test_gpu_serving import build_logprob  # noqa: E402


def _student_t(m, data, invcov):
    """A user likelihood with the reference's calling convention: ``m[1, nout]`` tensor, ``data[nout]``, ``invcov`` ->
    0-d tensor (util.py:953-955 is the default of this shape)."""
    d = m - data
    chi2 = (d @ invcov @ d.T)[0][0]
    return -0.5 * 5.0 * torch.log1p(chi2 / 4.0)


def test_user_loglikelihoodfunc_and_externalloglike():
    from linna_amd import util
    lp0, pred, yinv, prob = build_logprob("simple_6_4", 4.0)
    ext = lambda theta: -0.25 * float(np.sum(np.asarray(theta) ** 2))
    lp = util.Log_prob(lp0.data_new, lp0.invcov_new, pred, yinv, lp0.transform, 4.0, loglikelihoodfunc=_student_t,
                       externalloglike=ext)
    z = np.random.RandomState(3).standard_normal((40, 6)).astype(np.float32) * 0.5
    got = lp(z, returntorch=False)
    theta = np.atleast_2d(lp0.transform(z))
    m = yinv(pred.predict(torch.as_tensor(theta, dtype=torch.float32))).cpu().numpy().astype(np.float64)
    S = np.asarray(prob["invcov"], np.float64)
    d = m - np.asarray(prob["data"], np.float64)[None, :]
    chi2 = np.einsum("bi,ij,bj->b", d, S, d)
    ref = -2.5 * np.log1p(chi2 / 4.0) / 4.0 - 0.5 * np.sum(z.astype(np.float64) ** 2, axis=1) + np.array([ext(t) for t in theta])
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=2e-6)
    # one walker in, scalar out (util.py:990-1021), and the Gaussian default + externalloglike on the fused path
    assert np.ndim(lp(z[0], returntorch=False)) == 0
    lpg = util.Log_prob(lp0.data_new, lp0.invcov_new, pred, yinv, lp0.transform, 4.0, externalloglike=ext)
    base = lp0(z, returntorch=False)
    np.testing.assert_allclose(lpg(z, returntorch=False), base + np.array([ext(t) for t in theta], np.float32), rtol=2e-6, atol=2e-6)
    # a NaN from the user's function is a rejected point, not an error (util.py:1015-1016)
    lpn = util.Log_prob(lp0.data_new, lp0.invcov_new, pred, yinv, lp0.transform, 4.0, externalloglike=lambda th: float("nan"))
    assert np.all(np.isneginf(lpn(z[:4], returntorch=False)))


def test_callback_surface_matches_the_live_reference():
    """The same user ``loglikelihoodfunc`` (Student-t) and ``externalloglike`` handed to the LIVE reference's ``Log_prob``
    (util.py:990-1021), walker by walker at T = 4 (tests/golden/callbacks.npz): the host-callback path and the fused
    Gaussian path with an external term return the reference's numbers."""
    from linna_amd import util
    g = cases.golden("callbacks")
    lp0, pred, yinv, prob = build_logprob(str(g["case"]), 4.0)
    ext = lambda theta: -0.25 * float(np.sum(np.asarray(theta) ** 2))
    lps = util.Log_prob(lp0.data_new, lp0.invcov_new, pred, yinv, lp0.transform, 4.0, loglikelihoodfunc=_student_t, externalloglike=ext)
    lpg = util.Log_prob(lp0.data_new, lp0.invcov_new, pred, yinv, lp0.transform, 4.0, externalloglike=ext)
    np.testing.assert_allclose(lps(g["z"], returntorch=False), g["student"], rtol=3e-6, atol=3e-6)
    np.testing.assert_allclose(lpg(g["z"], returntorch=False), g["gauss_ext"], rtol=1.5e-5)
    np.testing.assert_allclose(float(lps(g["z"][7], returntorch=False)), g["student"][7], rtol=2e-6, atol=2e-6)   # one walker, scalar


class _Pool(object):
    """The surface ``ml_sampler_core`` uses of the reference's MPI pool (util.py:99-289)."""

    def __init__(self):
        self.noduplicate = False
        self.nmap = self.nitems = self.nclose = 0
        self.seen_noduplicate = []

    def map(self, fn, iterable):
        items = list(iterable)
        self.nmap += 1
        self.nitems += len(items)
        return [fn(x) for x in items]

    def is_master(self):
        return True

    def noduplicate_close(self):
        self.seen_noduplicate.append(self.noduplicate)
        self.nclose += 1
        self.noduplicate = False


def _problem2d(gauss=False):
    cov = np.diag([0.5, 0.2])
    means = np.array([0.1, 1.0])
    priors = [{"param": "test_%d" % i, "dist": "flat", "arg1": -2.0, "arg2": 2.0} for i in range(2)]
    if gauss:
        priors[1] = {"param": "test_1", "dist": "gauss", "arg1": 0.8, "arg2": 0.3}
    return means, cov, priors


def _theory(x, outdirs):
    return np.array(x[1], copy=True)


def _core(out, priors, means, cov, pool=None, params=None, method="emcee", nwalkers=8, **kw):
    from linna_amd.main import ml_sampler_core
    from linna_amd.nn import ChtoModelv2
    p = {"trainingoption": 1, "num_epochs": 40, "batch_size": 20}
    p.update(params or {})
    np.random.seed(0)
    torch.manual_seed(0)
    return ml_sampler_core([200, 200], [40, 40], [1, 1], [2, 2], [0.5, 0.5], [100, 100], [100, 100], out, _theory, priors, means, cov,
                           np.array([0.3, 0.6]), pool, nwalkers, "cuda", None, False, [2.0, 1.0], None, False, 1, None, ChtoModelv2,
                           p, method, **kw)


def test_pool_gauss_priors_importance_step_and_chisqcut(tmp_path):
    means, cov, priors = _problem2d(gauss=True)
    out = str(tmp_path) + "/run/"
    pool = _Pool()
    ext_calls = []

    def ext(theta):
        ext_calls.append(1)
        return 0.0
    chain, lp = _core(out, priors, means, cov, pool=pool, params={"nimp": 300}, chisqcut=1e3, loglikelihoodfunc=_student_t,
                      externalloglike=ext)
    # the pool evaluated the theory: 2 iterations x (train + val) + the importance step, nothing else went through it
    assert pool.nmap == 5 and pool.nclose == 2 and pool.seen_noduplicate == [True, True]
    assert pool.nitems <= 2 * 240 + 300 and pool.nitems >= 300 + 100
    assert len(ext_calls) > 0
    # with "nimp" the function returns the importance subsample and its emulator log-probabilities (main.py:311-335)
    assert chain.shape == (300, 2) and np.asarray(lp).shape == (300,)
    for f in ("samples_im.npy", "log_prob_samples_x.npy", "theory.npy", "weight_im.npy"):
        assert os.path.isfile(os.path.join(out, f)), f
    lps, logp, w = np.load(os.path.join(out, "weight_im.npy"))
    np.testing.assert_array_equal(np.load(os.path.join(out, "samples_im.npy")), chain)
    assert abs(w.sum() - 1.0) < 1e-9 and np.all(w >= 0) and (w == 0).sum() < 60
    # logp = -chi2/2 + log prior in theta space, through the dense log-likelihood kernel: against numpy in float64
    th = np.load(os.path.join(out, "theory.npy"))
    S = np.linalg.inv(cov)
    d = th - means[None, :]
    ref = -0.5 * np.einsum("bi,ij,bj->b", d, S, d) - 0.5 * (chain[:, 1] - 0.8) ** 2 / 0.3 ** 2
    np.testing.assert_allclose(logp, ref, rtol=2e-6, atol=2e-6)
    # gauss prior: theta_1 = 0.8 + 0.3 z, unbounded; flat prior: inside its box; the chain feels the prior (posterior of
    # theta_1: N(1, 0.2) x N(0.8, 0.09) -> mean 0.862)
    assert np.all(np.abs(chain[:, 0]) <= 2.0)
    # chisqcut dropped the training rows with y^T invcov y >= cut (util.py:1265) -- with 1e3 nothing is cut here
    x0 = np.loadtxt(os.path.join(out, "iter_0", "train_samples_x.txt"))
    assert x0.shape == (200, 2) and np.all(np.abs(x0[:, 0]) <= 2.0) and np.all(np.abs(x0[:, 1] - 0.8) <= 5 * 0.3)
    # second call: everything is read back from the artefacts
    chain2, lp2 = _core(out, priors, means, cov, pool=pool, params={"nimp": 300}, chisqcut=1e3, loglikelihoodfunc=_student_t,
                        externalloglike=ext)
    np.testing.assert_array_equal(chain, chain2)


def test_chisqcut_cuts_rows(tmp_path):
    from linna_amd import util
    rs = np.random.RandomState(2)
    for nout in (3, 33, 457):
        A = rs.standard_normal((nout, nout))
        S = A @ A.T / nout + np.eye(nout)
        y = rs.standard_normal((700, nout)) * rs.uniform(0.2, 3.0, size=(700, 1))
        x = rs.standard_normal((700, 4))
        ref = np.einsum("bi,ij,bj->b", y, S, y)
        np.testing.assert_allclose(util.chi2_rows_gpu(y, S), ref, rtol=6e-6)        # fp32 on the dense log-likelihood kernel
        np.testing.assert_allclose(util.chi2_rows(y, S), ref, rtol=1e-12)          # what the post steps use: float64, host
        fy, fx = str(tmp_path / "y.npy"), str(tmp_path / "x.txt")
        np.save(fy, y); np.savetxt(fx, x)
        cut = float(np.median(ref))
        util.chisqcut_all(None, S, cut, fy, fx)
        keep = ref < cut
        margin = np.abs(ref - cut) > 1e-3 * cut                     # rows at rounding distance of the cut may go either way
        got = np.load(fy)
        assert abs(len(got) - keep.sum()) <= (~margin).sum()
        assert np.loadtxt(fx).shape[0] == len(got)
    assert util.chi2_rows(np.zeros((0, 5)), np.eye(5)).shape == (0,) and util.chi2_rows_gpu(np.zeros((0, 5)), np.eye(5)).shape == (0,)


def test_importance_helpers_match_the_live_reference(tmp_path):
    """``logp_theory_data`` (util.py:1506-1517) and ``chisqcut_all`` (:1260-1270) (float64 quadratic forms on the host, as
    the reference's) against the LIVE reference's numbers on the same inputs
    (tests/golden/importance_helpers.npz): -chi2/2 + log prior of every sample (theory rows longer than the data vector
    cut as the reference cuts them; -inf outside a flat prior), and the rows a chi^2 cut keeps."""
    import cases
    from linna_amd import util
    g = cases.golden("importance_helpers")
    rs = np.random.RandomState(88)                                   # make_golden.importance_inputs
    nout, ndim, n = 7, 4, 60
    A = rs.standard_normal((nout, nout))
    cov = A @ A.T / nout + 0.3 * np.eye(nout)
    data = rs.uniform(0.5, 1.5, nout)
    theory = np.concatenate([data, [9.0, 9.0]])[None, :] + rs.standard_normal((n, nout + 2)) * 0.4
    samples = rs.uniform(-1.5, 1.5, (n, ndim))
    priors = [{"param": "a", "dist": "flat", "arg1": -1.0, "arg2": 1.2}, {"param": "b", "dist": "gauss", "arg1": 0.2, "arg2": 0.7},
              {"param": "c", "dist": "flat", "arg1": -1.4, "arg2": 1.4}, {"param": "d", "dist": "gauss", "arg1": -0.3, "arg2": 1.1}]
    invcov = np.linalg.inv(cov)
    logp = np.array(util.logp_theory_data(samples, theory, data, invcov, util.LogPrior(priors)), np.float64)
    np.testing.assert_array_equal(np.isinf(logp), np.isinf(g["logp"]))
    ok = np.isfinite(logp)
    np.testing.assert_allclose(logp[ok], g["logp"][ok], rtol=2e-6, atol=2e-6)
    fy, fx = str(tmp_path / "y.npy"), str(tmp_path / "x.txt")
    np.save(fy, theory[:, :nout]); np.savetxt(fx, samples)
    util.chisqcut_all(data, invcov, float(g["chisqcut"]), fy, fx)
    np.testing.assert_array_equal(np.load(fy), g["cut_y"])
    np.testing.assert_array_equal(np.loadtxt(fx), g["cut_x"])
