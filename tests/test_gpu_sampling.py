"""GPU: ensemble stretch move and HMC -- kernel-level replay against the oracle, the reference's
HMCSampler.py trace, posterior statistics on the 33-D Gaussian, and ml_sampler_core end to end."""
import os

import numpy as np
import pytest

import cases
from linna_amd import _lib

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
from test_gpu_serving import build_logprob  # noqa: E402


def identity_emulator_logprob(ndim, means, cov, priors, temperature=1.0):
    """An emulator that reproduces theory(theta) = theta EXACTLY: relu(x) - relu(-x) = x."""
    from linna_amd import nn, util, predictor_gpu
    m = nn.MLP(ndim, ndim, None, width=2 * ndim, depth=1)
    I = np.eye(ndim, dtype=np.float32)
    m.load_state_dict({"layer1.weight": np.concatenate([I, -I]), "layer1.bias": np.zeros(2 * ndim, np.float32),
                       "layer2.weight": np.concatenate([I, -I], axis=1), "layer2.bias": np.zeros(ndim, np.float32)})
    sigma = np.sqrt(np.diag(cov))
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32))
    pred = predictor_gpu.Predictor(ndim, ndim, model=m, device="cuda",
                                   X_transform=util.X_transform_class(t(np.zeros(ndim)), t(np.ones(ndim)), "cpu", None),
                                   y_transform=util.Y_transform_class(t(np.zeros(ndim)), t(1.0 / sigma), "cpu"))
    return util.Log_prob(t(means), t(np.linalg.inv(cov)), pred, util.Y_invtransform_data(sigma, "cpu"),
                         util.Transform(priors), temperature)


def test_stretch_kernels_replay_against_oracle():
    from oracle import sampling
    from linna_amd import sampler
    lp, pred, yinv, prob = build_logprob("simple_6_4", 4.0)
    nw, nd = 64, 6
    ens = sampler.EnsembleSampler(nw, nd, lp, seed=11, randomize_split=False)
    x0 = np.random.RandomState(1).standard_normal((nw, nd)).astype(np.float32) * 0.3
    ens.set_state(x0)
    logp0 = ens.logp.cpu().numpy().copy()
    ens.step()
    torch.cuda.synchronize()
    # replay: same Philox draws (seed as passed to the library, walker id, step 0, stream = half index)
    lib_seed = (11 + 0x9E3779B97F4A7C15 * 1) & 0xFFFFFFFFFFFFFFFF
    coords, logp = x0.copy(), logp0.copy()
    halves = np.arange(nw).reshape(2, nw // 2)
    from oracle import likelihood
    emu = cases.oracle_emulator(prob)
    f = lambda q: likelihood.log_prob(q, emu, prob["priors"], prob["data"], prob["invcov"], 4.0)
    for h in (0, 1):
        S, Cc = halves[h], halves[1 - h]
        u = sampling.walker_draws(lib_seed, 0, h, nw)[S]          # [ns, 4] uniforms for the active walkers
        bits = sampling.philox4x32(np.stack([S.astype(np.uint32), np.zeros(len(S), np.uint32), np.full(len(S), h, np.uint32),
                                             np.zeros(len(S), np.uint32)], 1),
                                   np.array([lib_seed & 0xFFFFFFFF, lib_seed >> 32], np.uint32))
        rint = ((bits[:, 1].astype(np.uint64) * np.uint64(len(Cc))) >> np.uint64(32)).astype(np.int64)
        q, fac = sampling.stretch_propose(coords[S], coords[Cc], u[:, 0], rint)
        new_lp = f(q)
        acc = sampling.stretch_accept(logp[S], new_lp, fac, u[:, 2])
        coords[S[acc]] = q[acc]
        logp[S[acc]] = new_lp[acc]
    got = ens.coords[:, :nd].cpu().numpy()
    # accept decisions can only differ where |lnpdiff - log u| is at rounding level; require near-total agreement
    same = np.all(np.abs(got - coords) <= 1e-5 * (1 + np.abs(coords)), axis=1)
    assert same.mean() >= 0.97, same.mean()
    np.testing.assert_allclose(ens.logp.cpu().numpy()[same], logp[same], rtol=8e-6)
    assert 0 < int(ens.naccept.sum()) < nw


def test_hmc_matches_reference_trace():
    """linna/HMCSampler.py driven by the reference Log_prob + autograd (golden hmc_trace), replayed
    through the product's HMCSampler with the same momentum / uniform draws."""
    from linna_amd import HMCSampler as H
    g = cases.golden("hmc_trace")
    lp = build_logprob(str(g["case"]), 1.0)[0]
    nin = g["x"].shape[1]
    s = H.HMCSampler(lp, np.zeros(nin, np.float32), np.ones(nin, np.float32))
    chain = s.sample(len(g["uniforms"]), int(g["num_steps"]), float(g["step_size"]), momenta=g["momenta"],
                     uniforms=g["uniforms"])
    acc = np.array([c["accepted"] for c in chain])
    assert (acc == g["accepted"]).all()
    np.testing.assert_allclose(np.stack([c["x"] for c in chain]), g["x"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose([c["lnP"] for c in chain], g["lnP"], rtol=4e-6)


def test_batched_hmc_step_matches_oracle():
    from oracle import sampling, likelihood
    from linna_amd import sampler
    lp, pred, yinv, prob = build_logprob("mlp_7_5_small", 16.0)
    emu = cases.oracle_emulator(prob)
    B, nd = 32, 7
    rs = np.random.RandomState(5)
    x0 = (0.2 * rs.standard_normal((B, nd))).astype(np.float32)
    p0 = rs.standard_normal((B, nd)).astype(np.float32)
    u = rs.uniform(size=B).astype(np.float32)
    mass = np.linspace(0.5, 2.0, nd).astype(np.float32)
    h = sampler.BatchedHMC(lp, x0, mass=mass)
    fg = lambda q: likelihood.grad_log_prob(q, emu, prob["priors"], prob["data"], prob["invcov"], 16.0)
    l0, g0 = fg(x0)
    np.testing.assert_allclose(h.lnp.cpu().numpy(), l0, rtol=1e-5)
    xn, ln, gn, acc = sampling.hmc_batched_step(fg, x0, l0, g0, mass, 4, 1e-3, p0, u)
    h.step(4, 1e-3, p0=p0, u=u)
    got_acc = h.naccept.cpu().numpy().astype(bool)
    assert (got_acc == acc).mean() >= 0.95
    ok = got_acc == acc
    np.testing.assert_allclose(h.x[:, :nd].cpu().numpy()[ok], xn[ok], rtol=2e-6, atol=2e-8)
    np.testing.assert_allclose(h.lnp.cpu().numpy()[ok], ln[ok], rtol=1e-5)


@pytest.mark.parametrize("name,B", [("mlp_33_33", 70), ("v2_33_33", 5), ("mlp_33_33_dense", 33), ("mlp_33_33", 2100)])
def test_leapfrog_in_the_gradient_launch_equals_the_separate_entries(name, B):
    """linna_hmc_start + linna_logprob_grad_leapfrog (momentum draw, first half kick and drift in one launch; every later kick
    and drift in the FINISH of the gradient launch) against linna_hmc_init / _kick_drift / linna_logprob_grad: the same
    arithmetic in the same order -> equal chains.  Plain MLP (GRAD program), ChtoModelv2 (forward + dX chain program), a dense
    covariance (no one-launch gradient: the kick follows as a launch of its own), and a batch the 16-row engine serves."""
    from linna_amd import sampler
    lp = build_logprob(name, 2.0)[0]
    nd = 33
    x0 = (0.2 * np.random.RandomState(B).standard_normal((B, nd))).astype(np.float32)
    mass = np.linspace(0.5, 2.0, nd).astype(np.float32)
    a = sampler.BatchedHMC(lp, x0, mass=mass, seed=3, fused=True)
    b = sampler.BatchedHMC(lp, x0, mass=mass, seed=3, fused=False)
    for it, (nleap, eps) in enumerate([(1, 1e-2), (5, 2e-2), (3, 5e-2), (4, 1e-2)]):
        a.step(nleap, eps); b.step(nleap, eps)
        torch.cuda.synchronize()
        for nm in ("x", "lnp", "g", "p", "q", "H0", "lnp_new", "g_new"):
            ta, tb = getattr(a, nm), getattr(b, nm)
            ta, tb = (ta[:, :nd], tb[:, :nd]) if ta.dim() == 2 else (ta, tb)
            assert torch.equal(ta, tb), (it, nm, float((ta - tb).abs().max()))
        assert torch.equal(a.naccept, b.naccept)
    assert 0 < int(a.naccept.sum()) <= 4 * B


def test_batched_hmc_proposals_match_the_references_integrator():
    """The per-walker HMC move of the reference (sampler.py:59-98 ``_hmc_wrapper``, :141-143 the Metropolis test; code it
    never reaches through emcee, called directly for tests/golden/hmc_move.npz with the autograd gradient of its own
    ``Log_prob``): ``BatchedHMC`` with the same momenta proposes the same points, and with given uniforms accepts exactly
    where ``log u < lnP(q) - lnP(x) + factor`` of the reference's numbers says so."""
    from linna_amd import sampler
    g = cases.golden("hmc_move")
    lp = build_logprob(str(g["case"]), 1.0)[0]
    x0 = g["coords"].astype(np.float32)
    nw, nd = x0.shape
    mass = g["var"].astype(np.float32)
    p0 = (g["momenta"] / np.sqrt(g["var"])[None, :]).astype(np.float32)
    h = sampler.BatchedHMC(lp, x0, mass=mass)
    np.testing.assert_allclose(h.lnp.cpu().numpy(), g["lnp_old"], rtol=5e-6)
    h.step(int(g["nsteps"]), float(g["epsilon"]), p0=p0, u=np.zeros(nw, np.float32))       # u = 0: every finite proposal is taken
    assert h.naccept.cpu().numpy().all()
    np.testing.assert_allclose(h.x[:, :nd].cpu().numpy(), g["q"], rtol=3e-6, atol=3e-7)
    np.testing.assert_allclose(h.lnp.cpu().numpy(), g["lnp_new"], rtol=5e-6)
    # the acceptance rule on the reference's own numbers
    u = np.random.RandomState(3).uniform(size=nw).astype(np.float32)
    want = np.log(u) < (g["lnp_new"] - g["lnp_old"] + g["factor"])
    margin = np.abs(np.log(u) - (g["lnp_new"] - g["lnp_old"] + g["factor"])) > 0.05      # (float32 energies: skip the knife edges)
    h2 = sampler.BatchedHMC(lp, x0, mass=mass)
    h2.step(int(g["nsteps"]), float(g["epsilon"]), p0=p0, u=u)
    got = h2.naccept.cpu().numpy().astype(bool)
    assert margin.sum() >= nw - 4 and (got[margin] == want[margin]).all()
    assert 0 < want.sum() < nw                                                              # both outcomes occur


def _gaussian_33():
    rs = np.random.RandomState(0)                                    # README.rst:69-80 shaped
    ndim = 33
    means = rs.uniform(size=ndim)
    cov = np.diag(0.1 * rs.uniform(0.2, 1.0, size=ndim))
    priors = [{"param": "test_%d" % i, "dist": "flat", "arg1": -5.0, "arg2": 5.0} for i in range(ndim)]
    return ndim, means, cov, priors


def test_hip_proposals_satisfy_the_fixture_relation():
    """The relation every transition of the reference-held emcee chain satisfies (tests/test_stretch_fixture.py,
    tests/stretch_relation.py) holds for what the HIP kernels do to the same 4-walker, 2-parameter ensembles: the
    proposals of ``linna_stretch_propose`` lie on the line through the walker and one walker of the complementary half
    with zz in [1/2, 2] and factor (ndim - 1) log zz, and whole iterations of the one-launch half steps are red/blue
    stretch steps of the stored ensemble."""
    import ctypes as C
    from linna_amd import sampler, _lib
    from linna_amd.sampler import ChainStore
    from stretch_relation import explain_step, partners
    d = ChainStore.read_h5(os.path.join(cases.GOLDEN, "2dgaussian_Fulltconn/iter_0/chemcee_256.h5"))
    chain = np.asarray(d["chain"], np.float32)
    nw, nd = chain.shape[1], chain.shape[2]
    means, cov = np.array([0.1, 1.0]), np.diag([0.5, 0.2])
    priors = [{"param": "p%d" % i, "dist": "flat", "arg1": -2.0, "arg2": 2.0} for i in range(nd)]
    lp = identity_emulator_logprob(nd, means, cov, priors)
    ens = sampler.EnsembleSampler(nw, nd, lp, seed=21)
    ctx, st = _lib.ctx(ens.dev.index), _lib.stream()
    S = torch.tensor([0, 2], dtype=torch.int32, device="cuda")
    Cc = torch.tensor([1, 3], dtype=torch.int32, device="cuda")
    zzs = []
    for t in range(0, 200, 5):                                           # 40 stored ensembles x 2 proposals
        ens.set_state(chain[t])
        ens.step_dev.fill_(t)
        _lib.call("linna_stretch_propose", ctx, _lib.ptr(ens.coords), ens.ld, nd, _lib.iptr(S), 2, _lib.ptr(ens.coords), ens.ld,
                  _lib.iptr(Cc), 2, C.c_uint64(77), _lib.iptr(ens.step_dev), 0, 2.0, _lib.ptr(ens.Q), ens.ld, _lib.ptr(ens.factors), st)
        q, fac = ens.Q[:, :nd].cpu().numpy().astype(np.float64), ens.factors.cpu().numpy().astype(np.float64)
        for i, k in enumerate((0, 2)):
            p = partners(chain[t, k].astype(np.float64), q[i], [(j, chain[t, j].astype(np.float64)) for j in (1, 3)], tol=2e-6)
            assert p, (t, k)
            np.testing.assert_allclose(fac[i], (nd - 1) * np.log(p[0][1]), atol=3e-6)
            zzs.append(p[0][1])
    zzs = np.array(zzs)
    assert zzs.min() >= 0.5 - 1e-6 and zzs.max() <= 2.0 + 1e-6 and 0.25 < np.mean(zzs < 1.0) < 0.75
    # whole iterations (two fused half steps each, random equal splits) from a stored ensemble
    ens = sampler.EnsembleSampler(nw, nd, lp, seed=22)
    ens.set_state(chain[100])
    prev, nmoved = chain[100].astype(np.float64), 0
    for it in range(60):
        ens.step()
        cur = ens.coords[:, :nd].cpu().numpy().astype(np.float64)
        ex = explain_step(prev, cur, tol=3e-6)
        assert ex is not None, it
        nmoved += len(ex[1])
        prev = cur
    assert ens.fused and 60 < nmoved < 240


@pytest.mark.parametrize("name,nw", [("mlp_33_33", 200), ("v2_33_33", 64), ("custom_7_5", 34)])
def test_fused_half_step_is_bit_identical_to_three_launches(name, nw):
    """linna_stretch_half_step (propose + whole-network log-probability + accept in one launch) against
    linna_stretch_propose / linna_logprob_eval / linna_stretch_accept: same Philox counters, same
    arithmetic, so the states must be EQUAL, ragged last workgroup included."""
    from linna_amd import sampler
    from test_gpu_serving import _custom_problem
    custom = _custom_problem(7, 5, 31, 48, 3) if name == "custom_7_5" else None
    lp, pred, yinv, prob = build_logprob(name, 2.0, prob=custom)
    nd = prob["nin"]
    x0 = (0.3 * np.random.RandomState(5).standard_normal((nw, nd))).astype(np.float32)
    a = sampler.EnsembleSampler(nw, nd, lp, seed=21)
    b = sampler.EnsembleSampler(nw, nd, lp, seed=21, fused=False)
    a.set_state(x0); b.set_state(x0)
    for _ in range(6):
        a.step(); b.step()
    torch.cuda.synchronize()
    assert a.fused is True and b.fused is False
    assert torch.equal(a.coords, b.coords)
    assert torch.equal(a.logp, b.logp)
    assert torch.equal(a.naccept, b.naccept)
    assert 0 < int(a.naccept.sum()) < 6 * nw


def test_ensemble_posterior_33d_gaussian():
    """BASELINE target: posterior mean within 0.05 sigma on the 33-D Gaussian (here against the
    ANALYTIC posterior, which is what the reference CPU path samples for theory = identity)."""
    from linna_amd import sampler, util
    ndim, means, cov, priors = _gaussian_33()
    lp = identity_emulator_logprob(ndim, means, cov, priors)
    nw = 2048
    ens = sampler.EnsembleSampler(nw, ndim, lp, seed=3)
    z0 = util.invTransform(priors)(means)[None, :] + 0.01 * np.random.RandomState(1).standard_normal((nw, ndim))
    ens.set_state(z0)
    ens.run(1500, store=False)
    c, l = ens.run(600)
    th = ens.theta_of(c).cpu().numpy().reshape(-1, ndim)
    sig = np.sqrt(np.diag(cov))
    assert np.max(np.abs(th.mean(0) - means) / sig) < 0.05
    np.testing.assert_allclose(th.std(0), sig, rtol=0.06)
    acc = float(ens.naccept.float().mean()) / ens.iteration
    assert 0.05 < acc < 0.6


def test_hmc_posterior_33d_gaussian():
    from linna_amd import sampler, util
    ndim, means, cov, priors = _gaussian_33()
    lp = identity_emulator_logprob(ndim, means, cov, priors)
    B = 1024
    z0 = util.invTransform(priors)(means)[None, :] + 0.01 * np.random.RandomState(2).standard_normal((B, ndim))
    h = sampler.BatchedHMC(lp, z0.astype(np.float32), seed=9)
    h.sample(100, 5, 0.004)
    chain, lnps = h.sample(300, 5, 0.004)
    th = sampler.EnsembleSampler(2, ndim, lp).theta_of(chain).cpu().numpy().reshape(-1, ndim)
    sig = np.sqrt(np.diag(cov))
    acc = float(h.naccept.float().mean()) / 400
    assert acc > 0.5, acc
    assert np.max(np.abs(th.mean(0) - means) / sig) < 0.05
    np.testing.assert_allclose(th.std(0), sig, rtol=0.08)


def test_ml_sampler_core_end_to_end(tmp_path):
    """tests/test_main.py:43-45 of the reference (``testmain``): 2-D Gaussian, one iteration,
    20 training points, 10 epochs, batch 5, emcee, 4 walkers, theory = identity."""
    from linna_amd.main import ml_sampler_core
    from linna_amd.nn import ChtoModelv2
    from copy import deepcopy
    np.random.seed(0)
    ndim = 2
    init = np.random.uniform(size=ndim)
    cov = np.diag([0.5, 0.2])
    means = np.array([0.1, 1])
    priors = [{"param": "test_%d" % i, "dist": "flat", "arg1": -2.0, "arg2": 2.0} for i in range(ndim)]

    def theory(x, outdirs):
        return deepcopy(x[1])

    params = {"trainingoption": 1, "num_epochs": 10, "batch_size": 5}
    out = str(tmp_path) + "/2dgaussian/"
    chain, logprob = ml_sampler_core([20], [5], [1], [2], [0.5], [100], [100], out, theory, priors, means, cov, init, None, 4,
                                     "cuda", None, False, [1.0], omegab2cut=None, docuda=False, tsize=1, gpunode=None,
                                     nnmodel_in=ChtoModelv2, params=params, method="emcee")
    assert chain.ndim == 2 and chain.shape[1] == ndim and len(chain) > 0
    assert np.all(np.isfinite(chain)) and np.all(np.abs(chain) <= 2.0)
    for f in ("train_samples_x.txt", "train_samples_y.npy", "best.pth.tar", "X_transform.pkl", "finish.pkl", "lr.npy",
              "chemcee_256.h5", "model_args.pkl"):
        assert os.path.isfile(os.path.join(out, "iter_0", f)), f
    # second call: every stage is skipped because its artefact exists (tests/test_main.py:47-51)
    chain2, _ = ml_sampler_core([20], [5], [1], [2], [0.5], [100], [100], out, theory, priors, means, cov, init, None, 4, "cuda",
                                None, False, [1.0], nnmodel_in=ChtoModelv2, params=params, method="emcee")
    np.testing.assert_array_equal(chain, chain2)


def test_slice_ensemble_posterior_33d_gaussian():
    """zeus-style ensemble slice sampler: posterior of the 33-D Gaussian and mu tuning."""
    from linna_amd import sampler, util
    ndim, means, cov, priors = _gaussian_33()
    lp = identity_emulator_logprob(ndim, means, cov, priors)
    nw = 512
    ens = sampler.SliceEnsembleSampler(nw, ndim, lp, seed=5)
    z0 = util.invTransform(priors)(means)[None, :] + 0.001 * np.random.RandomState(1).standard_normal((nw, ndim))
    ens.set_state(z0)
    ens.run(300, store=False)                               # the walkers start in a 1e-3 ball: let them spread
    assert not ens.tune and 0.05 < ens.mu < 50.0            # tuning converged
    c, l = ens.run(500)
    th = ens.theta_of(c).cpu().numpy().reshape(-1, ndim)
    sig = np.sqrt(np.diag(cov))
    assert np.max(np.abs(th.mean(0) - means) / sig) < 0.05
    np.testing.assert_allclose(th.std(0), sig, rtol=0.06)
    # every stored log-probability is the log-probability of the stored position
    z_last = c[-1]
    np.testing.assert_allclose(lp.evaluate(torch.nn.functional.pad(z_last, (0, ens.ld - ndim))).cpu().numpy(),
                               l[-1].cpu().numpy(), rtol=1e-4, atol=1e-4)
    assert ens._fast_ok is True                             # the tuned iterations ran through linna_slice_half_step
    assert ens.neval / (ens.iteration * nw) < 40            # speculative rounds (8 bracket ends per side, 16 trials), inactive lanes included


def test_slice_engine_choice_and_overflow_fallback_keep_the_posterior():
    """with the engines of the later rounds left AUTOMATIC (chosen from the usage counters at fixed iterations) a
    walker's stored lnP and its trial lnP may come from different engines (last-bit differences): the chain is then no longer
    bit-equal to the ``USE_EXPECT = False`` chain -- bit equality across the switch holds per engine only
    (test_slice_later_rounds_follow_the_usage_counters) -- but it must sample the same posterior: means within the
    Monte-Carlo error of each other and of the truth.  And a run whose one-call path overflows mid-run (a schedule one
    stepping-out step deep) is redone on the round loop, deepens its schedule and goes on: still the posterior."""
    from linna_amd import sampler, util
    ndim, means, cov, priors = _gaussian_33()
    lp = identity_emulator_logprob(ndim, means, cov, priors)
    sig = np.sqrt(np.diag(cov))
    nw = 600
    z0 = util.invTransform(priors)(means)[None, :] + 0.001 * np.random.RandomState(2).standard_normal((nw, ndim))
    out = {}
    try:
        for use in (True, False):
            sampler.SliceEnsembleSampler.USE_EXPECT = use
            ens = sampler.SliceEnsembleSampler(nw, ndim, lp, seed=7)
            ens.set_state(z0)
            ens.run(300, store=False)
            c, _ = ens.run(400)
            th = ens.theta_of(c).cpu().numpy().reshape(-1, ndim)
            out[use] = (th.mean(0), th.std(0), ens.expected_rows)
            assert ens._fast_ok is True and not ens.tune
    finally:
        sampler.SliceEnsembleSampler.USE_EXPECT = True
    assert out[True][2] is not None and out[False][2] is None       # the automatic choice was live in one run only
    for use in (True, False):
        assert np.max(np.abs(out[use][0] - means) / sig) < 0.06
        np.testing.assert_allclose(out[use][1], sig, rtol=0.07)
    assert np.max(np.abs(out[True][0] - out[False][0]) / sig) < 0.08  # two independent estimates of the same mean
    # overflow mid-run: one stepping-out step per side is not enough for every walker
    ens = sampler.SliceEnsembleSampler(nw, ndim, lp, seed=8)
    ens.set_state(z0)
    ens.run(300, store=False)
    ens.set_schedule([1], [4, 8])
    c, l = ens.run(300)
    assert ens.noverflow >= 1 and getattr(ens, "_esc_level", 0) >= 1 and sum(ens.m_sched) > 1     # redone, and looking further ahead now
    th = ens.theta_of(c).cpu().numpy().reshape(-1, ndim)
    assert np.max(np.abs(th.mean(0) - means) / sig) < 0.08
    np.testing.assert_allclose(th.std(0), sig, rtol=0.08)
    np.testing.assert_allclose(lp.evaluate(torch.nn.functional.pad(c[-1], (0, ens.ld - ndim))).cpu().numpy(), l[-1].cpu().numpy(),
                               rtol=1e-4, atol=1e-4)


def test_zeus_driver_smoke(tmp_path):
    """tests/test_sampler.py:4-14 of the reference (``test_zeus``) on the emulator path."""
    from linna_amd import sampler, util
    ndim, means, cov, priors = 2, np.array([0.1, 1.0]), np.diag([0.5, 0.2]), \
        [{"param": "t%d" % i, "dist": "flat", "arg1": -5.0, "arg2": 5.0} for i in range(2)]
    lp = identity_emulator_logprob(ndim, means, cov, priors)
    x0 = util.invTransform(priors)(means)[None, :] + 0.001 * np.random.RandomState(3).standard_normal((10, ndim))
    samp = sampler.ZeusSampler(lp, ndim, 10, x0=x0, transform=util.Transform(priors))
    store = samp.sample(None, 1000, outdir=str(tmp_path), ntimes=10, tautol=0.01, incremental=True)
    z, th, l = store.arrays()
    assert th.shape[1:] == (10, 2) and len(th) >= 100 and np.all(np.isfinite(l))
    assert os.path.isfile(os.path.join(str(tmp_path), "zeus_256.h5"))
    assert np.all(np.abs(th.reshape(-1, 2).mean(0) - means) < 0.25)


def test_hessian_of_log_prob_matches_oracle_fd():
    """util.py:1037-1051 intended semantics (Ddlnp): Hessian of lnP wrt z."""
    from oracle import likelihood
    from linna_amd import util
    lp, pred, yinv, prob = build_logprob("mlp_7_5_small", 16.0)
    dd = util.Ddlnp(prob["data"], prob["invcov"], pred, yinv, util.Transform(prob["priors"]), 16.0)
    z0 = np.array([0.1, -0.2, 0.3, 0.05, -0.1, 0.2, 0.0], np.float32)
    H = dd(z0)
    emu = cases.oracle_emulator(prob)
    eps, n = 1e-2, 7
    E = eps * np.eye(n)
    _, gp = likelihood.grad_log_prob(z0[None, :] + E, emu, prob["priors"], prob["data"], prob["invcov"], 16.0, dtype=np.float64)
    _, gm = likelihood.grad_log_prob(z0[None, :] - E, emu, prob["priors"], prob["data"], prob["invcov"], 16.0, dtype=np.float64)
    Href = (gp - gm) / (2 * eps)
    Href = 0.5 * (Href + Href.T)
    assert H.shape == (n, n)
    np.testing.assert_allclose(H, Href, rtol=5e-6, atol=5e-6 * np.abs(Href).max())
    g = util.Dlnp(prob["data"], prob["invcov"], pred, yinv, util.Transform(prob["priors"]), 16.0)(z0)
    _, gref = likelihood.grad_log_prob(z0[None, :], emu, prob["priors"], prob["data"], prob["invcov"], 16.0)
    np.testing.assert_allclose(g, gref[0], rtol=4e-5, atol=1.5e-6 * np.abs(gref).max())


def test_device_convergence_statistics_match_host_estimators():
    """DeviceChain.integrated_time / checkmeanstd (GPU, batched FFT over walkers) against the host restatements
    of emcee's estimator and sampler.py:370-387 on an AR(1) chain with known autocorrelation."""
    from linna_amd import sampler
    rs = np.random.RandomState(4)
    nt, nw, nd = 600, 24, 3
    rho = np.array([0.5, 0.8, 0.9])
    x = np.zeros((nt, nw, nd))
    e = rs.standard_normal((nt, nw, nd))
    for i in range(1, nt):
        x[i] = rho * x[i - 1] + e[i]
    dc = sampler.DeviceChain()
    for blk in np.array_split(x.astype(np.float32), 6):       # arrives in blocks, as in the driver loop
        dc.append(torch.as_tensor(blk, device="cuda"))
    xf = x.astype(np.float32).astype(np.float64)
    np.testing.assert_allclose(dc.integrated_time(), sampler.integrated_time(xf), rtol=1e-8)
    np.testing.assert_allclose(dc.integrated_time(discard=120), sampler.integrated_time(xf[120:]), rtol=1e-8)
    np.testing.assert_allclose(dc.integrated_time(upto=450), sampler.integrated_time(xf[:450]), rtol=1e-8)
    np.testing.assert_allclose(dc.integrated_time(discard=90, upto=450), sampler.integrated_time(xf[90:450]), rtol=1e-8)
    assert np.all(np.abs(dc.integrated_time() / ((1 + rho) / (1 - rho)) - 1) < 0.5)    # AR(1): tau = (1+rho)/(1-rho)
    for n in (40, 333):
        assert dc.checkmeanstd(n, 0.3, 0.3) == bool(sampler.checkmeanstd(xf[-n:], 0.3, 0.3))
    assert dc.last(250).shape == (250, nw, nd) and torch.equal(dc.last(250).cpu(), torch.as_tensor(x[-250:].astype(np.float32)))


def test_posterior_matches_cpu_oracle_chain():
    """The GPU pipeline (fused stretch half steps around the whole-network kernel) and a CPU chain driven by the
    oracle's stretch move + the oracle's log-probability sample the posterior of the SAME emulator (a 7 -> 48 x 3
    -> 5 MLP with random trained-shape weights, flat and Gaussian priors).  7-D: the autocorrelation time is
    100-200 iterations and both chains hold thousands of independent samples: the 5/25/50/75/95 % quantiles of
    every parameter within 0.08 sigma, correlation matrices within 0.06."""
    from oracle import likelihood, sampling
    from linna_amd import sampler
    from test_gpu_serving import _custom_problem
    prob = _custom_problem(7, 5, 31, 48, 3)
    lp, pred, yinv, _ = build_logprob(None, 1.0, prob=prob)
    emu = cases.oracle_emulator(prob)
    f = lambda q: likelihood.log_prob(q.astype(np.float32), emu, prob["priors"], prob["data"], prob["invcov"], 1.0)
    nd, nburn = 7, 1500
    rs = np.random.RandomState(12)
    # ---- CPU: oracle stretch move, numpy RNG, 64 walkers
    nw = 64
    x0 = (0.3 * rs.standard_normal((nw, nd))).astype(np.float32)
    coords, logp = x0.astype(np.float64).copy(), f(x0)
    halves = np.arange(nw).reshape(2, nw // 2)
    keep = []
    for it in range(nburn + 10000):
        for h in (0, 1):
            S, Cc = halves[h], halves[1 - h]
            coords, logp, _ = sampling.stretch_half_step(coords, logp, S, Cc, rs.uniform(size=len(S)), rs.randint(0, len(Cc), len(S)),
                                                         rs.uniform(size=len(S)), f)
        if it >= nburn:
            keep.append(coords.copy())
    cpu = np.concatenate(keep)
    # ---- GPU: 1024 walkers
    nwg = 1024
    ens = sampler.EnsembleSampler(nwg, nd, lp, seed=5)
    ens.set_state((0.3 * rs.standard_normal((nwg, nd))).astype(np.float32))
    ens.run(nburn, store=False)
    c, _ = ens.run(4000)
    assert ens.fused is True
    gpu = c.cpu().numpy().reshape(-1, nd)
    tau = sampler.integrated_time(np.stack(keep)[:, :16, :])
    assert np.all(tau < 400), tau                                  # >= 25 autocorrelation times per walker on the CPU side
    # robust summaries: this posterior has a faint far mode in the erf tail of a flat prior where a single CPU walker
    # can sit for thousands of iterations (seen with one seed: 0.1 % quantile at -3 sigma_z, std inflated by 20 %)
    qs = [0.05, 0.25, 0.5, 0.75, 0.95]
    qc, qg = np.quantile(cpu, qs, axis=0), np.quantile(gpu, qs, axis=0)
    sd = (qc[4] - qc[0]) / 3.29                                   # robust sigma from the central 90 %
    assert np.max(np.abs(qg - qc) / sd) < 0.08, np.abs(qg - qc) / sd
    core = lambda a: a[np.all(np.abs(a - qc[2]) < 5 * sd, axis=1)]
    assert np.abs(np.corrcoef(core(gpu).T) - np.corrcoef(core(cpu).T)).max() < 0.06


def test_slice_sampler_fused_trial_points_are_bit_identical():
    """linna_logprob_eval_slice_points (trial points built in the kernel's prologue, never written) against
    linna_slice_points + linna_logprob_eval_if: the same arithmetic, so the chains must be EQUAL."""
    from linna_amd import sampler
    lp, pred, yinv, prob = build_logprob("mlp_33_33", 2.0)
    nw, nd = 96, 33
    x0 = (0.3 * np.random.RandomState(8).standard_normal((nw, nd))).astype(np.float32)
    a = sampler.SliceEnsembleSampler(nw, nd, lp, seed=4)
    b = sampler.SliceEnsembleSampler(nw, nd, lp, seed=4)
    b.fused_points = False
    a.set_state(x0); b.set_state(x0)
    for _ in range(5):
        a.step(); b.step()
    torch.cuda.synchronize()
    assert a.fused_points is True and b.fused_points is False
    assert torch.equal(a.coords, b.coords) and torch.equal(a.logp, b.logp)
    assert a.mu == b.mu and a.neval == b.neval


@pytest.mark.parametrize("nw,m_sched,nt_sched", [(34, [1, 1, 2, 3, 8], [1, 3, 7, 21]), (250, [3, 5, 5], [5, 2, 9, 16]),
                                                 (2050, [1, 2, 4, 8], [2, 4, 8, 16, 32]), (6, [2, 16], [33]),
                                                 # ONE stepping-out round: its logic rides in the first shrinking round (linna_slice_fusion)
                                                 (34, [8], [16, 16]), (250, [4], [8, 3]), (130, [16], [32]), (6, [2], [3, 9]), (1026, [3], [5, 7, 30])])
def test_one_call_slice_schedules_and_ragged_ensembles(nw, m_sched, nt_sched):
    """The rounds after the first evaluate only the trial points of the walkers still active (a device-side list and count,
    read by the evaluation's prologue): whatever the schedule of bracket ends / trials per round and the ensemble size
    (half ensembles of 17, 125, 1025 and 3 walkers; schedules that shrink, grow, are odd), the chain is the round loop's,
    bit for bit, and so are the expansion / contraction counts."""
    from linna_amd import sampler
    lp, pred, yinv, prob = build_logprob("mlp_33_33", 2.0)
    nd = 33
    x0 = (0.3 * np.random.RandomState(nw).standard_normal((nw, nd))).astype(np.float32)
    _lib.engine_rows(4)
    try:
        a = sampler.SliceEnsembleSampler(nw, nd, lp, seed=9, tune=False, mu=0.9, fast=True)
        b = sampler.SliceEnsembleSampler(nw, nd, lp, seed=9, tune=False, mu=0.9, fast=False)
        a.set_schedule(m_sched, nt_sched)
        a.set_state(x0); b.set_state(x0)
        for it in range(6):
            a.step(); b.step()
            torch.cuda.synchronize()
            assert torch.equal(a.coords, b.coords) and torch.equal(a.logp, b.logp), it
            if a.noverflow:                              # (a short schedule left a walker unfinished: the guarded step went back,
                break                                    #  redid the iteration on the round loop and deepened the schedule: new buffers)
            ca = a._fast_bufs["counters"].cpu().numpy()
            cb = b.counters.cpu().numpy()
            assert ca[0] == cb[0] and ca[1] == cb[1], (it, ca[:4], cb)
        assert a._fast_ok is True
        # evaluated points: every walker of a half ensemble in the first round of each kind, then only the listed ones
        ns = nw // 2
        per_it = 2 * (2 * m_sched[0] + nt_sched[0]) * ns
        assert a.neval >= (it + 1) * per_it - 1 or a.noverflow
        assert a.neval <= (it + 1) * 2 * (2 * sum(m_sched) + sum(nt_sched)) * ns or a.noverflow
    finally:
        _lib.engine_rows(0)


@pytest.mark.parametrize("rows", [4, 8, 16])
def test_slice_fusion_masks_give_the_same_chain(rows):
    """linna_slice_fusion: with one stepping-out round the half step drops launches by folding their work into the
    neighbouring ones (the trial points of the first shrinking round derived in the evaluation's prologue from the bracket
    ends' lnP; ...).  Every mask must give the chain of mask 0, bit for bit, with the same expansion / contraction / point
    counts -- on each engine of the whole-network kernel (the prologue differs per engine)."""
    from linna_amd import sampler
    lp, pred, yinv, prob = build_logprob("mlp_33_33", 2.0)
    nd = 33
    prev = _lib.slice_fusion(-1)
    _lib.engine_rows(rows)
    try:
        for nw, sched in [(128, ([8], [16, 16])), (44, ([5], [9])), (600, ([2], [4, 8]))]:
            x0 = (0.3 * np.random.RandomState(nw).standard_normal((nw, nd))).astype(np.float32)
            out = {}
            for mask in (0, 1, 3, 7):
                _lib.slice_fusion(mask)
                a = sampler.SliceEnsembleSampler(nw, nd, lp, seed=21, tune=False, mu=0.8, fast=True)
                a.set_schedule(*sched)
                a.set_state(x0)
                for it in range(7):
                    a._step()                              # (no guard: an unfinished walker is part of the comparison)
                torch.cuda.synchronize()
                assert a._fast_ok is True
                out[mask] = (a.coords.clone(), a.logp.clone(), a._fast_bufs["counters"][:4].cpu().numpy(), int(a.step_dev.item()))
            for mask in (1, 3, 7):
                assert torch.equal(out[mask][0], out[0][0]) and torch.equal(out[mask][1], out[0][1]), (nw, mask)
                assert (out[mask][2] == out[0][2]).all() and out[mask][3] == out[0][3], (nw, mask, out[mask][2], out[0][2])
    finally:
        _lib.engine_rows(0)
        _lib.slice_fusion(prev)


def test_slice_later_rounds_follow_the_usage_counters():
    """The launches of the rounds after the first run the engine that suits the number of trial points the usage counters
    have shown (``expect_rows`` of linna_slice_half_step, refreshed at one-call iterations 16, 64, ...): with ONE engine forced
    for every launch the expectation cannot change a bit of the chain -- same coordinates with and without it -- and after 70
    iterations it holds sane numbers: at most every point of a round, 2^20 for a round that has practically never run."""
    from linna_amd import sampler
    lp, pred, yinv, prob = build_logprob("mlp_33_33", 2.0)
    nw, nd = 600, 33
    x0 = (0.3 * np.random.RandomState(3).standard_normal((nw, nd))).astype(np.float32)
    _lib.engine_rows(4)
    try:
        out = {}
        for use in (True, False):
            sampler.SliceEnsembleSampler.USE_EXPECT = use
            a = sampler.SliceEnsembleSampler(nw, nd, lp, seed=5, tune=False, mu=0.8, fast=True)
            a.set_state(x0)
            a.run(70, store=False)
            torch.cuda.synchronize()
            assert a._fast_ok is True and a.noverflow == 0
            out[use] = (a.coords.clone(), a.logp.clone(), a.expected_rows)
        assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1])
        rows = out[True][2]
        assert out[False][2] is None and rows is not None and len(rows) == a.nexp_rounds + a.nshr_rounds
        pts = [2 * m for m in a.m_sched] + list(a.nt_sched)
        for i, r in enumerate(rows):
            if i in (0, a.nexp_rounds):
                assert r == 1                                   # (the first round of each kind evaluates every walker: entry unused)
            else:
                assert r == 1 << 20 or 8 <= r <= pts[i] * a.half * 1.25 + 8
        assert rows[a.nexp_rounds + 1] < 1 << 20               # the second shrinking round does run (a few % of the walkers)
    finally:
        sampler.SliceEnsembleSampler.USE_EXPECT = True
        _lib.engine_rows(0)


@pytest.mark.parametrize("name,nw", [("mlp_33_33", 96), ("v2_33_33", 16), ("mlp_33_33", 1024)])
def test_one_call_slice_half_step_equals_the_round_loop(name, nw):
    """linna_slice_half_step (speculative rounds: several bracket ends / trials per evaluation launch, a fixed launch
    sequence gated on the device, no host wait) against the round-by-round loop over linna_slice_init / _expand / _draw /
    _shrink / _commit: same Philox counters, same comparisons -> the chains must be EQUAL, as must the expansion and
    contraction counts that tune mu.  Ensembles of 16, 96 and 1024 walkers (first rounds of 8 / 8 / 4 ends per side and
    16 / 16 / 8 trials; the later rounds evaluate the walkers still active only, listed on the device, and look further)."""
    from linna_amd import sampler
    lp, pred, yinv, prob = build_logprob(name, 2.0)
    nd = 33
    x0 = (0.3 * np.random.RandomState(8).standard_normal((nw, nd))).astype(np.float32)
    _lib.engine_rows(4)                                     # one engine for every batch size: bit-equal evaluations
    try:
        a = sampler.SliceEnsembleSampler(nw, nd, lp, seed=4, tune=False, mu=0.7, fast=True)
        b = sampler.SliceEnsembleSampler(nw, nd, lp, seed=4, tune=False, mu=0.7, fast=False)
        a.set_state(x0); b.set_state(x0)
        for it in range(12):
            a.step(); b.step()
            torch.cuda.synchronize()
            assert torch.equal(a.coords, b.coords) and torch.equal(a.logp, b.logp), it
            ca = a._fast_bufs["counters"].cpu().numpy()
            cb = b.counters.cpu().numpy()
            assert ca[0] == cb[0] and ca[1] == cb[1] and ca[2] == 0, (it, ca[:4], cb)      # expansions, contractions, none unfinished
    finally:
        _lib.engine_rows(0)
    assert a._fast_ok is True and a.iteration == b.iteration == 12
    assert a.neval > 0 and b.neval > 0
    assert a.nexp_rounds + a.nshr_rounds <= 9               # (the rounds after the first look further ahead each time)
    # a walker that cannot finish within the call's rounds: the sampler goes back to where the run started and redoes it on
    # the round loop -- the chain is the round loop's (here: one expansion round of one end per side, mu too small)
    c = sampler.SliceEnsembleSampler(nw, nd, lp, seed=4, tune=False, mu=0.05, fast=True)
    d = sampler.SliceEnsembleSampler(nw, nd, lp, seed=4, tune=False, mu=0.05, fast=False)
    c.set_schedule([1], c.nt_sched)
    c.set_state(x0); d.set_state(x0)
    cc, cl = c.run(3)
    dc, dl = d.run(3)
    assert c.noverflow == 1 and d.noverflow == 0
    assert torch.equal(cc, dc) and torch.equal(cl, dl) and torch.equal(c.coords, d.coords) and c.iteration == d.iteration == 3
    assert int(c.step_dev.item()) == int(d.step_dev.item())
