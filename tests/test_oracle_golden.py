"""CPU: pin the numpy oracle against the golden vectors produced by the live reference."""
import numpy as np
import pytest

import cases
import synth
from oracle import emulator, likelihood, training, sampling


@pytest.mark.parametrize("name", [c[0] for c in cases.SERVING])
def test_serving_matches_reference(name):
    g = cases.golden(name)
    prob = cases.serving_problem(name)
    emu = cases.oracle_emulator(prob)
    z = g["z"]
    theta = likelihood.prior_map(z, prob["priors"])
    np.testing.assert_allclose(theta, g["theta"], rtol=2e-6, atol=2e-6)
    m = emu.predict(theta)
    scale = np.abs(g["m"]).max()
    np.testing.assert_allclose(m, g["m"], rtol=2e-4, atol=2e-5 * scale)
    for j, T in enumerate(g["temps"]):
        ll = likelihood.log_prob(z, emu, prob["priors"], prob["data"], prob["invcov"], float(T))
        np.testing.assert_allclose(ll, g["loglike"][:, j], rtol=5e-4)
    # fp64 evaluation of the same chain agrees with the fp32 reference to fp32 accuracy
    ll64 = likelihood.log_prob(z, emu, prob["priors"], prob["data"], prob["invcov"], 1.0, dtype=np.float64)
    np.testing.assert_allclose(ll64, g["loglike"][:, 0], rtol=5e-4)


@pytest.mark.parametrize("name", [c[0] for c in cases.SERVING])
def test_grad_log_prob_matches_autograd(name):
    g = cases.golden(name)
    prob = cases.serving_problem(name)
    emu = cases.oracle_emulator(prob)
    lnp, grad = likelihood.grad_log_prob(g["z"], emu, prob["priors"], prob["data"], prob["invcov"], 1.0)
    np.testing.assert_allclose(lnp, g["loglike"][:, 0], rtol=5e-4)
    scale = np.abs(g["grad"]).max(axis=1, keepdims=True)
    assert np.all(np.abs(grad - g["grad"]) <= 2e-3 * scale + 1e-5)


def test_per_walker_equals_batched():
    prob = cases.serving_problem("simple_6_4")
    emu = cases.oracle_emulator(prob)
    z = cases.golden("simple_6_4")["z"][:8]
    a = likelihood.log_prob(z, emu, prob["priors"], prob["data"], prob["invcov"], 4.0)
    b = likelihood.log_prob_per_walker(z, emu, prob["priors"], prob["data"], prob["invcov"], 4.0)
    np.testing.assert_allclose(a, b, rtol=1e-5)


def test_fixture_2d_known_answers():
    """the reference fixture's log-probabilities (constants only; the checkpoint
    itself is read by the product loader in test_host_api)."""
    g = cases.golden("fixture2d")
    np.testing.assert_allclose(g["loglike"][:4], [-2.9208457, -3.2061472, -4.2610073, -4.2800837], rtol=1e-6)
    np.testing.assert_allclose(g["sigma"], np.sqrt([0.5, 0.2]), rtol=1e-6)


@pytest.mark.parametrize("name", [c[0] for c in cases.TRAIN])
def test_training_step_matches_reference(name):
    g = cases.golden(name)
    p = cases.training_problem(name)
    icov = training.normalised_inverse_cov(p["cov"], p["sigma"], p["y_std"])
    np.testing.assert_allclose(icov, g["icov_norm"], rtol=1e-5, atol=1e-6 * np.abs(g["icov_norm"]).max())
    data_norm = training.normalise_target(p["data"][None, :], p["sigma"], p["y_mean"], p["y_std"])[0]
    np.testing.assert_allclose(data_norm, g["data_norm"].reshape(-1), rtol=1e-5, atol=1e-6)
    stats = dict(X_mean=p["X_mean"], X_std=p["X_std"], y_mean=p["y_mean"], y_std=p["y_std"],
                 sigma=p["sigma"].astype(np.float32), data_norm=g["data_norm"].reshape(-1), icov_norm=g["icov_norm"])
    params = {k: v.copy() for k, v in p["weights"].items()}
    # forward + loss pieces on minibatch 0
    x = (p["X"][0] - p["X_mean"][None, :]) / p["X_std"][None, :]
    pred = emulator.forward(params, x, p["kind"], p["nin"], p["nout"], **p["kw"])
    np.testing.assert_allclose(pred, g["pred0"], rtol=1e-3, atol=1e-4 * np.abs(g["pred0"]).max())
    lrow, cMd, cnnd, _, _ = training.aux(g["pred0"], p["Y"][0], stats["data_norm"], stats["icov_norm"],
                                         stats["sigma"], p["y_mean"], p["y_std"])
    np.testing.assert_allclose(lrow, g["loss_rows0"], rtol=2e-4)
    np.testing.assert_allclose(cMd, g["chisqMd0"], rtol=2e-4)
    np.testing.assert_allclose(cnnd, g["chisqnnd0"], rtol=2e-4)
    vm = training.val_metric(g["pred0"], p["Y"][0], stats["data_norm"], stats["icov_norm"], stats["sigma"],
                             p["y_mean"], p["y_std"])
    np.testing.assert_allclose(vm, g["val0"], rtol=2e-4)
    l, dpred = training.loss_grad(g["pred0"], p["Y"][0], stats["data_norm"], stats["icov_norm"],
                                  stats["sigma"], p["y_mean"], p["y_std"])
    np.testing.assert_allclose(dpred, g["dpred0"], rtol=1e-3, atol=1e-5 * np.abs(g["dpred0"]).max())
    # three optimiser steps
    opt = training.new_opt_state(params)
    losses = []
    for s in range(3):
        l, grads = training.train_step(params, opt, p["X"][s], p["Y"][s], stats, p["kind"], p["nin"], p["nout"],
                                       lr=float(g["lr"]), **p["kw"])
        losses.append(l)
        if s == 0:
            for k, gk in grads.items():
                ref = g["grad0/" + k]
                got = gk if p["full"] else synth.tensor_digest(gk)
                np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-5 * np.abs(ref).max() + 1e-9, err_msg=k)
        for k, v in params.items():
            ref = g["param%d/%s" % (s + 1, k)]
            got = v if p["full"] else synth.tensor_digest(v)
            np.testing.assert_allclose(got, ref, rtol=1e-3, atol=2e-4 * np.abs(ref).max() + 1e-7, err_msg=k)
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-3)


def test_data_statistics_match_train_NN():
    g = cases.golden("train_nn_run")
    sigma = np.sqrt(np.diag(g["cov"]))
    X_mean, X_std, y_mean, y_std = training.data_statistics(g["train_x"], g["train_y"], g["train_y"], sigma)
    np.testing.assert_allclose(X_mean, g["X_mean"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(X_std, g["X_std"], rtol=1e-5)
    np.testing.assert_allclose(y_mean, g["y_mean"], rtol=1e-6)
    np.testing.assert_allclose(y_std, g["y_std"], rtol=1e-5)


def test_ypositive_statistics_and_loss_constants_match_train_NN():
    """``ypositive=True`` (util.py:1410-1431, 1444-1447, 567-586) against the live reference's train_NN: rows dropped, the
    log-space median / MAD, the normalised data vector and the inverse of log(1 + E C E) in the normalised space; then the
    oracle's training step reproduces the first epoch's per-step losses from the reference's initial weights."""
    g = cases.golden("train_nn_ypos")
    sigma = np.sqrt(np.diag(g["cov"]))
    tx, ty, tyl, vx, vy = training.ypositive_clip(g["train_x"], g["train_y"], g["train_y"], g["val_x"], g["val_y"])
    assert len(tx) == len(ty) == int(g["ntrain"]) == 200 and len(vx) == len(vy) == int(g["nval"]) == 50
    assert ty.max() == 1e10 and ty.min() == 1e-30
    X_mean, X_std, y_mean, y_std = training.data_statistics(tx, ty, tyl, sigma, ypositive=True)
    np.testing.assert_allclose(X_mean, g["X_mean"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(X_std, g["X_std"], rtol=1e-5)
    np.testing.assert_allclose(y_mean, g["y_mean"], rtol=2e-6)
    np.testing.assert_allclose(y_std, g["y_std"], rtol=1e-5)
    dn = training.normalise_data(g["data"], sigma, y_mean, y_std, ypositive=True)
    np.testing.assert_allclose(dn, g["data_norm"].reshape(-1), rtol=2e-5, atol=2e-6)
    ic = training.normalised_inverse_cov(g["cov"], sigma, y_std, ypositive=True, data=g["data"])
    np.testing.assert_allclose(ic, g["icov_norm"], rtol=2e-4, atol=2e-5 * np.abs(g["icov_norm"]).max())
    # first epoch: torch.manual_seed(1234) + DataLoader(shuffle) order of 200 rows in batches of 50 (drop_last)
    import torch
    from linna_amd import predictor_gpu, util
    torch.manual_seed(1234)                                          # predictor_gpu.py:221
    order = predictor_gpu.BatchLoader(util.ArrayDataset(tx, ty), 50, shuffle=True, drop_last=True).epoch_order().numpy()
    stats = dict(X_mean=g["X_mean"], X_std=g["X_std"], y_mean=g["y_mean"], y_std=g["y_std"], sigma=sigma.astype(np.float32),
                 data_norm=g["data_norm"].reshape(-1), icov_norm=g["icov_norm"], ypositive=True)
    params = {k: v.copy() for k, v in synth.weights("ChtoModelv2", 5, 4, 311).items()}
    opt = training.new_opt_state(params)
    losses = []
    for b in range(4):
        rows = order[b * 50:(b + 1) * 50]
        l, _ = training.train_step(params, opt, tx[rows], ty[rows].astype(np.float32), stats, "ChtoModelv2", 5, 4, float(g["lr"]))
        losses.append(float(l))
    np.testing.assert_allclose(losses, g["train_losses"][:4], rtol=2e-4)


def test_hmc_chain_matches_reference_trace():
    g = cases.golden("hmc_trace")
    name = str(g["case"])
    prob = cases.serving_problem(name)
    emu = cases.oracle_emulator(prob)

    def f(x):
        l, gr = likelihood.grad_log_prob(x[None, :], emu, prob["priors"], prob["data"], prob["invcov"], 1.0)
        return l[0], gr[0]

    xs, lnps, acc = sampling.hmc_chain(f, np.zeros(prob["nin"], np.float32), np.ones(prob["nin"], np.float32),
                                       len(g["uniforms"]), int(g["num_steps"]), float(g["step_size"]),
                                       g["momenta"], g["uniforms"])
    assert (acc == g["accepted"]).all()
    theta = likelihood.prior_map(xs, prob["priors"])       # HMCSampler.py:60 stores transform(x); identity here
    np.testing.assert_allclose(xs, g["x"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(lnps, g["lnP"], rtol=1e-3)
    assert theta.shape == xs.shape


def test_philox_known_answer():
    """Random123 kat_vectors: philox4x32-10, counter = key = 0 and all-ones."""
    out0 = sampling.philox4x32(np.zeros(4, np.uint32), np.zeros(2, np.uint32))
    assert [hex(int(v)) for v in out0] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    ones = np.full(4, 0xFFFFFFFF, np.uint32)
    out1 = sampling.philox4x32(ones, ones[:2])
    assert [hex(int(v)) for v in out1] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]


def test_stretch_move_detailed_balance_gaussian():
    """Unpinned restatement: check statistically on an analytic 5-D Gaussian."""
    rs = np.random.RandomState(0)
    nd, nw = 5, 64
    var = np.linspace(0.5, 2.0, nd).astype(np.float32)
    f = lambda q: (-0.5 * (q * q / var).sum(-1)).astype(np.float32)
    x = rs.standard_normal((nw, nd)).astype(np.float32)
    lp = f(x)
    keep = []
    for it in range(3000):
        inds = rs.permutation(nw) % 2
        for split in (0, 1):
            S = np.where(inds == split)[0]
            C = np.where(inds != split)[0]
            x, lp, _ = sampling.stretch_half_step(x, lp, S, C, rs.uniform(size=len(S)).astype(np.float32),
                                                  rs.randint(len(C), size=len(S)), rs.uniform(size=len(S)), f)
        if it > 500:
            keep.append(x.copy())
    s = np.concatenate(keep)
    assert np.all(np.abs(s.mean(0)) < 0.08)
    np.testing.assert_allclose(s.var(0), var, rtol=0.08)


def test_per_walker_hmc_move_matches_the_references_integrator():
    """sampler.py:59-98 (``_hmc_wrapper``: the per-walker leapfrog of the HMC move the reference wrote but never reaches
    through emcee) called directly in the live reference with the autograd gradient of its own ``Log_prob``
    (tests/golden/hmc_move.npz): the oracle's batched step proposes the same points with the same kinetic-energy factor."""
    from oracle import sampling
    g = cases.golden("hmc_move")
    prob = cases.serving_problem(str(g["case"]))
    emu = cases.oracle_emulator(prob)
    fg = lambda q: likelihood.grad_log_prob(q, emu, prob["priors"], prob["data"], prob["invcov"], 1.0)
    x0 = g["coords"].astype(np.float32)
    l0, g0 = fg(x0)
    np.testing.assert_allclose(l0, g["lnp_old"], rtol=2e-4)
    mass = g["var"].astype(np.float32)
    p0 = (g["momenta"] / np.sqrt(g["var"])[None, :]).astype(np.float32)
    det = {}
    sampling.hmc_batched_step(fg, x0, l0, g0, mass, int(g["nsteps"]), float(g["epsilon"]), p0, np.zeros(len(x0), np.float32), details=det)
    np.testing.assert_allclose(det["q"], g["q"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(det["lnp_new"], g["lnp_new"], rtol=5e-4)
    np.testing.assert_allclose(det["factor"], g["factor"], rtol=5e-3, atol=5e-3)


def test_oracle_under_ill_conditioned_covariances():
    """tests/golden/cond_26_457.npz (make_golden.py ``cond``, live reference): dense covariances of condition 1e2 / 1e4 /
    1e6 with residuals that are draws from the covariance at the anchors.  The oracle's emulator reproduces the reference's
    m; its float64 log-probability equals the float64 value stored with the golden (same fp32-rounded inverse covariance)
    to 1e-9, and the reference's own fp32 value sits within the summation error this condition number allows -- the
    numbers tests/test_gpu_cond.py holds the HIP path to."""
    g = cases.golden("cond_26_457")
    nin, nout, seed = 26, 457, 110
    X_mean, X_std, y_mean, y_std = synth.transform_constants(nin, nout, seed)
    _, _, priors = synth.gaussian_problem(nin, nout, seed, dense=False)
    w = synth.weights("ChtoModelv2", nin, nout, seed)
    worst = {}
    for ci, cond in enumerate(g["conds"]):
        cov, inv, half = synth.cond_problem(nin, nout, seed, float(cond))
        emu = likelihood.Emulator("ChtoModelv2", nin, nout, w, X_mean, X_std, y_mean, y_std, np.sqrt(np.diag(cov)))
        inv32 = inv.astype(np.float32)
        for k in range(g["z"].shape[0]):
            z = g["z"][k]
            m = emu.predict(likelihood.prior_map(z, priors))
            scale = np.abs(g["m/%d" % ci][k]).max()
            np.testing.assert_allclose(m, g["m/%d" % ci][k], rtol=2e-4, atol=2e-5 * scale)
            # the anchor's residual is the stored draw: data = m(z0) - cov^(1/2) xi
            np.testing.assert_allclose(g["m/%d" % ci][k][0].astype(np.float64) - g["data/%d" % ci][k], half @ g["xi"][k], atol=1e-9)
            d = g["m/%d" % ci][k].astype(np.float64) - g["data/%d" % ci][k].astype(np.float32).astype(np.float64)
            l64 = -0.5 * np.einsum("bi,ij,bj->b", d, inv32.astype(np.float64), d) - 0.5 * (z.astype(np.float64) ** 2).sum(-1)
            np.testing.assert_allclose(l64, g["lnP64/%d" % ci][k], rtol=1e-9)
            worst[float(cond)] = max(worst.get(float(cond), 0.0), float(np.abs(g["lnP32/%d" % ci][k] - l64)[0]))
    # the reference's own fp32 error at the anchors (chi^2 ~ nout) grows with the condition number
    assert worst[1e2] < 1e-3 and worst[1e4] < 1e-2 and 1e-3 < worst[1e6] < 0.5, worst
