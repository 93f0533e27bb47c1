"""GPU: inference-level parity for the reference's OWN network class -- ``ml_sampler`` hard-wires ``ChtoModelv2``
(main.py:70; nn.py:59-133) and the CosmoLike driver runs it with ``dolog10index=[0, 1]`` (cosmolike_run.py:184,320).

BASELINE configs[2] shape: 26 parameters -> 457 data points (nout >> nin, where ``relu(Linear(500, nout))`` ->
``Linear(nout, nout)`` is no bottleneck, unlike the README's 33 -> 33).  ``theory(theta) = A theta + c`` with a dense
SPD covariance and noise-free data, so the posterior is Gaussian in closed form: mean = theta_true, covariance =
(A^T Sigma^-1 A)^-1 (tests/golden/linear457.py; neighbouring parameters are made degenerate: correlations up to 0.69).
Parameters 0 and 1 have positive flat priors and reach the emulator as log10 -- the X-transform path of the whole-network
kernel -- so the emulated map is NOT linear in its inputs.  Flat priors are > 8 sigma away.

The whole loop of ``ml_sampler_core`` with ``ml_sampler``'s emcee schedule (four iterations, T = 16, 4, 1, 1): Latin
hypercube, theory callback, ``train_NN`` (LR range test, early stopping controller), checkpoint round trip, dense
log-likelihood in the one-launch ensemble half step, chain -> next design.
"""
import numpy as np
import pytest
import torch

import linear457

pytestmark = pytest.mark.gpu

NWALKERS = 4096
MEAN_TOL = 0.05          # max |posterior mean - truth| / sigma over the 26 parameters (north_star: 0.05 sigma)
STD_TOL = 0.05           # max |std / sigma - 1|
CORR_TOL = 0.05          # max |corr - corr_exact| over the 325 pairs (Monte-Carlo error ~0.01 at ~16 k independent samples)
LNP_MEDIAN_TOL = 0.1     # |median(stored lnP - exact lnP)|: emulator + fp32 dense quadratic form against float64 numpy
LNP_P99_TOL = 0.8        # 99th percentile of |stored lnP - exact lnP| (measured at 2048 walkers: median 0.006, p99 0.2)


def test_ml_sampler_core_posterior_through_chtomodelv2_26_457(tmp_path):
    from linna_amd.main import ml_sampler_core
    from linna_amd import nn, util
    prob = linear457.problem()
    out = str(tmp_path) + "/v2/"
    np.random.seed(0)
    torch.manual_seed(linear457.SEED)
    params = {"trainingoption": 1, "num_epochs": 400, "batch_size": 500}
    chain, logp = ml_sampler_core([10000] * 4, [500] * 4, [2, 2, 5, 4], [5, 5, 10, 15], [0.03, 0.03, 0.02, 0.01], [0.2] * 4,
                                  [0.15] * 4, out, linear457.Theory(prob["A"], prob["c"]), prob["priors"], prob["data"], prob["cov"],
                                  prob["init"], None, NWALKERS, "cuda", [0, 1], False, [4.0, 2.0, 1.0, 1.0], None, False, 1, None,
                                  nn.ChtoModelv2, params, "emcee")
    assert chain.ndim == 2 and chain.shape[1] == linear457.NIN and len(chain) > 400000 and np.all(np.isfinite(chain))
    bias, std_err, corr_err = linear457.summary(chain, prob)
    print("ChtoModelv2(26,457): bias %.4f sigma, std %.4f, corr %.4f, %d samples" % (bias, std_err, corr_err, len(chain)))
    assert bias < MEAN_TOL, bias
    assert std_err < STD_TOL, std_err
    assert corr_err < CORR_TOL, corr_err
    # the network the chain came through is the reference's class, with the log10 columns in its input transform
    model, _ = util.retrieve_model(out + "iter_3/", linear457.NIN, linear457.NOUT, nn.ChtoModelv2)
    assert type(model.model).__name__ == "ChtoModelv2" and list(model.X_transform.dolog10index) == [0, 1]
    # stored lnP (emulator, fp32, dense Sigma^-1 inside the one-launch half step) against the exact posterior in float64
    lp = np.asarray(logp).reshape(-1)[-len(chain):]
    sub = np.random.RandomState(1).randint(0, len(chain), 20000)
    z = np.asarray(util.invTransform(prob["priors"])(chain[sub]))
    d = chain[sub] @ prob["A"].T + prob["c"] - prob["data"]
    exact = -0.5 * np.einsum("bi,ij,bj->b", d, prob["icov"], d) - 0.5 * np.sum(z ** 2, axis=1)
    err = lp[sub] - exact
    print("stored lnP - exact: median %.3f, p99 |.| %.3f" % (np.median(err), np.percentile(np.abs(err), 99)))
    assert abs(np.median(err)) < LNP_MEDIAN_TOL and np.percentile(np.abs(err), 99) < LNP_P99_TOL
    import shutil
    shutil.rmtree(out, ignore_errors=True)              # (the four chain files are 1-4 GB each)
