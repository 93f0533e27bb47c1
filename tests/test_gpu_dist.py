"""The multi-rank paths of the sampler and of the trainer on device tensors -- gradient all-reduce,
complementary-ensemble all-gather, chain gather -- against the single-rank results they must reproduce.

Transport: with two devices visible every rank takes its own GPU and the collectives are the library's RCCL entries
(``linna_comm_init`` / ``linna_allreduce_sum_f32`` / ``linna_allgather_f32``); on the ONE-GPU test box the two ranks
share device 0 and torch.distributed's gloo carries them (RCCL refuses two ranks on one device) -- there RCCL is
exercised by the single-rank self-test through the C ABI below."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    two = torch.cuda.device_count() >= world
    torch.cuda.set_device(rank if two else 0)
    from linna_amd import dist as ldist
    # the product's own bring-up: rendezvous (gloo here), then -- one rank per device only -- the RCCL communicator through
    # the C ABI with its bounded self-test and the MIN-agreement over the ranks (linna_amd.dist.init)
    assert ldist.init(backend="gloo", device=torch.device("cuda", rank if two else 0), comm=two) == world
    try:
        assert ldist.collectives() is not None and (("RCCL" in ldist.collectives()) == two)
        ret[rank] = fn(rank, world)
    finally:
        ldist.shutdown()


def _run(fn, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _engine(name, B, world):
    import synth  # noqa: F401  (tests/golden on sys.path through conftest)
    from linna_amd import nn, util, predictor_gpu, trainer
    p = cases.training_problem(name)
    cls = {"ChtoModelv2": nn.ChtoModelv2, "MLP": nn.MLP}[p["kind"]]
    model = cls(p["nin"], p["nout"], None, **p["kw"])
    model.load_state_dict(p["weights"])
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32))
    pred = predictor_gpu.Predictor(p["nin"], p["nout"], model=model, device="cuda",
                                   X_transform=util.X_transform_class(t(p["X_mean"]), t(p["X_std"]), "cpu", None),
                                   y_transform=util.Y_transform_class(t(p["y_mean"]), t(p["y_std"]), "cpu"))
    X = p["X"].reshape(-1, p["nin"]); Y = p["Y"].reshape(-1, p["nout"])
    ytd = util.Y_transform_data(p["sigma"], "cpu")
    yinv = util.Y_invtransform_class(t(p["y_mean"]), t(p["y_std"]), t(p["data"]), "cpu")
    lf = util.Loss_fn(t(p["data"]), torch.tensor(p["cov"], dtype=torch.float64),
                      torch.tensor(np.linalg.inv(p["cov"]), dtype=torch.float64), ytd, yinv, "cpu")
    loader = predictor_gpu.BatchLoader(util.ArrayDataset(X, Y), B, shuffle=False, drop_last=True)
    eng = trainer.TrainEngine(pred, loader, lf, None, world_size=world, dist_group=None)
    return p, model, eng


def _train_job(rank, world):
    from linna_amd.predictor_gpu import _AdamWState
    name = "train_v2_12_40"
    Bg = 48                                                  # global batch; each rank takes half of it
    p, model, eng = _engine(name, Bg // world, world)
    opt = _AdamWState(model, 1e-3 * world, weight_decay=1e-4)        # lr * size, predictor_gpu.py:246
    for s in range(3):
        rows = torch.arange(s * Bg + rank * (Bg // world), s * Bg + (rank + 1) * (Bg // world), dtype=torch.int32, device="cuda")
        eng.step(opt, rows)
    torch.cuda.synchronize()
    return model.flat_params().cpu().numpy(), float(eng.loss_mean.item())


def test_data_parallel_training_matches_the_global_batch():
    """2 ranks x 24 rows with one gradient all-reduce per step == 1 rank x 48 rows (same lr * size)."""
    from linna_amd.predictor_gpu import _AdamWState
    res = _run(_train_job)
    np.testing.assert_array_equal(res[0][0], res[1][0])              # identical replicas after identical updates
    p, model, eng = _engine("train_v2_12_40", 48, 1)
    opt = _AdamWState(model, 2e-3, weight_decay=1e-4)
    for s in range(3):
        eng.step(opt, torch.arange(s * 48, (s + 1) * 48, dtype=torch.int32, device="cuda"))
    torch.cuda.synchronize()
    ref = model.flat_params().cpu().numpy()
    np.testing.assert_allclose(res[0][0], ref, rtol=2e-4, atol=1e-5)      # (fp32 reassociation: 2 x 24 rows + all-reduce against 48 rows)
    np.testing.assert_allclose(res[0][1], float(eng.loss_mean.item()), rtol=2e-6)


def test_rccl_single_rank_through_the_c_abi():
    """linna_comm_unique_id -> linna_comm_init (one rank) -> all-reduce / all-gather / broadcast on the caller's stream
    -> linna_comm_destroy, and the same through linna_amd.dist's helpers that the trainer and the sampler call."""
    import ctypes as C
    from linna_amd import _lib, dist as ldist
    dev = torch.cuda.current_device()
    assert ldist.comm_info(dev)[1] == 0                              # no communicator yet
    with pytest.raises(_lib.LinnaHipError):
        _lib.call("linna_allreduce_sum_f32", _lib.ctx(dev), _lib.ptr(torch.ones(4, device="cuda")), 4, _lib.stream())
    r, n = ldist.comm_init(dev, rank=0, world=1)
    try:
        assert (r, n) == (0, 1)
        rank, nranks, ver = ldist.comm_info(dev)
        assert (rank, nranks) == (0, 1) and ver > 20000              # RCCL 2.x reports 2xxyy
        assert ldist.comm_selftest(dev, timeout=30.0)                   # what bench.py asks before it commits to the communicator
        g = torch.arange(1000003, dtype=torch.float32, device="cuda")
        ref = g.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                                # on the caller's stream, whichever it is
            _lib.call("linna_allreduce_sum_f32", _lib.ctx(dev), _lib.ptr(g), g.numel(), _lib.stream())
            out = torch.empty(2 * 77, dtype=torch.float32, device="cuda")
            _lib.call("linna_allgather_f32", _lib.ctx(dev), _lib.ptr(g[:154].contiguous()), _lib.ptr(out), 154, _lib.stream())
            _lib.call("linna_broadcast_f32", _lib.ctx(dev), _lib.ptr(out), 154, 0, _lib.stream())
        side.synchronize()
        assert torch.equal(g, ref) and torch.equal(out, ref[:154])
        with pytest.raises(_lib.LinnaHipError):
            ldist._lib.call("linna_comm_init", _lib.ctx(dev), 0, 1, C.create_string_buffer(128))   # one per context
        # the helpers: gradient + loss scalar in one call when the scalar sits behind the buffer
        from linna_amd import nn
        m = nn.MLP(5, 3, None, width=16, depth=2).to("cuda")
        fg, tail = m.flat_grads(), m.grad_tail()
        fg.fill_(2.0); tail.fill_(7.0)
        assert tail.data_ptr() == fg.data_ptr() + 4 * fg.numel() and ldist.comm_active(fg)
        ldist.allreduce_grads(fg, tail)
        torch.cuda.synchronize()
        assert float(fg.sum()) == 2.0 * fg.numel() and float(tail) == 7.0
        assert ldist.broadcast_value(0.125, device="cuda:%d" % dev) == 0.125
    finally:
        ldist.comm_destroy(dev)
    assert ldist.comm_info(dev)[1] == 0


def _trainer_run_job(rank, world):
    """``Predictor.train`` on two ranks with lr.npy absent (rank 0 runs the range test alone, the value is broadcast)
    and a short patience (every rank must leave the epoch loop at the same epoch)."""
    import tempfile
    from linna_amd import util, nn, predictor_gpu
    g = cases.golden("train_nn_run")
    tmp = os.environ["LINNA_TEST_SHARED_DIR"] + "/"
    cov = g["cov"]
    sigma = np.sqrt(np.diag(cov))
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32))
    Xt = util.X_transform_class(t(g["X_mean"]), t(g["X_std"]), "cpu", None)
    Yt = util.Y_transform_class(t(g["y_mean"]), t(g["y_std"]), "cpu")
    ytd = util.Y_transform_data(sigma, "cpu")
    yinv = util.Y_invtransform_class(t(g["y_mean"]), t(g["y_std"]), t(g["data"]), "cpu")
    lf = util.Loss_fn(t(g["data"]), torch.tensor(cov, dtype=torch.float64), torch.tensor(np.linalg.inv(cov), dtype=torch.float64),
                      ytd, yinv, "cpu")
    torch.manual_seed(3)
    model = nn.ChtoModelv2(g["train_x"].shape[1], g["train_y"].shape[1], None)
    pred = predictor_gpu.Predictor(model.in_size, model.out_size, model=model, device="cuda", optim="automatic",
                                   X_transform=Xt, y_transform=Yt, outdir=tmp)
    loader = predictor_gpu.BatchLoader(util.ArrayDataset(g["train_x"], g["train_y"]), 25, shuffle=True, drop_last=True)
    vloader = predictor_gpu.BatchLoader(util.ArrayDataset(g["val_x"], g["val_y"]), len(g["val_y"]), shuffle=False)
    tl, vm = pred.train(loader, 400, lf, vloader, lf, initfrombest=False, rank=rank, size=world, patience=8)
    torch.cuda.synchronize()
    return len(vm), float(np.load(tmp + "lr.npy")), model.flat_params().cpu().numpy(), float(pred.optim.lr)


def test_trainer_run_on_two_ranks_range_test_and_early_stop(tmp_path, monkeypatch):
    monkeypatch.setenv("LINNA_TEST_SHARED_DIR", str(tmp_path))
    res = _run(_trainer_run_job)
    assert res[0][0] == res[1][0] and 8 < res[0][0] < 400            # both ranks stopped, at the same epoch, early
    assert 1e-4 <= res[0][1] <= 5e-3 and res[0][1] == res[1][1]      # one range test, one value
    np.testing.assert_array_equal(res[0][2], res[1][2])              # identical replicas throughout
    assert res[0][3] == res[1][3]


def _sampler_job(rank, world):
    from linna_amd import sampler, util, dist as ldist
    from test_gpu_sampling import identity_emulator_logprob, _gaussian_33
    ndim, means, cov, priors = _gaussian_33()
    lp = identity_emulator_logprob(ndim, means, cov, priors)
    nw = 128
    ens = sampler.EnsembleSampler(nw, ndim, lp, seed=3, dist_group=None, exchange="allgather")
    z0 = util.invTransform(priors)(means)[None, :] + 0.01 * np.random.RandomState(10 + rank).standard_normal((nw, ndim))
    ens.set_state(z0)
    ens.run(600, store=False)
    c, l = ens.run(300)
    allc, alll = ldist.gather_chain(c, l)
    th = ens.theta_of(allc).cpu().numpy()
    return th.reshape(-1, ndim).mean(0), th.reshape(-1, ndim).std(0), allc.shape, float(c.mean()), bool(ens.fused)


def test_ensemble_with_complement_exchange_and_chain_gather():
    """2 ranks x 128 walkers drawing stretch partners from the walkers of BOTH ranks (one all-gather per half step),
    chain gathered to every rank: the 33-D Gaussian posterior, and both ranks hold the same gathered chain."""
    from test_gpu_sampling import _gaussian_33
    ndim, means, cov, priors = _gaussian_33()
    res = _run(_sampler_job)
    sig = np.sqrt(np.diag(cov))
    for r in res:
        assert r[2] == (300, 256, ndim) and r[4]
        assert np.max(np.abs(r[0] - means) / sig) < 0.15             # 300 steps x 256 walkers, tau ~ 50
        np.testing.assert_allclose(r[1], sig, rtol=0.2)
    np.testing.assert_array_equal(res[0][0], res[1][0])              # the gathered chain is the same on both ranks
    assert res[0][3] != res[1][3]                                    # ... while their own walkers differ


def _driver_job(rank, world):
    import contextlib
    import io
    from linna_amd import sampler, util
    from test_gpu_sampling import identity_emulator_logprob, _gaussian_33
    ndim, means, cov, priors = _gaussian_33()
    lp = identity_emulator_logprob(ndim, means, cov, priors)
    out = os.environ["LINNA_TEST_SHARED_DIR"]
    nw = 256                                                          # walkers of the ONE ensemble: 128 per rank
    x0 = util.invTransform(priors)(means)[None, :] + 0.01 * np.random.RandomState(10 + rank).standard_normal((nw, ndim))
    drv = sampler.HMCSampler(lp, None, None, ndim, nw, x0=x0, transform=util.Transform(priors), seed=5,
                             exchange=os.environ.get("LINNA_TEST_EXCHANGE") or None)
    with contextlib.redirect_stdout(io.StringIO()) as log:
        drv.sample(None, 1000, outdir=out, ntimes=1e9, tautol=1e-9, incremental=True)      # never "converged": 1000 iterations
    if os.environ.get("LINNA_TEST_EXCHANGE") != "allgather":
        assert drv.exchange is None
    d = sampler.ChainStore.load(os.path.join(out, "chemcee_256.h5"))                        # every rank reads rank 0's file
    th = np.asarray(d["chain_transformed"])[600:]
    acc = np.asarray(d["accepted"])
    assert acc.shape[-1] == nw and (acc.reshape(-1, nw)[-1] > 0).all()     # acceptance counts of ALL walkers, rank order
    if rank == 0:
        # the statistic the stop rule reads is the one-rank statistic of the file's chain: tau over ALL 256 walkers
        import re
        from oracle import sampling as osamp
        checks = re.findall(r"max tau, ninter: \S+, \S+, (\S+), (\d+)", log.getvalue())
        tau_max, n_last = float(checks[-1][0]), int(checks[-1][1])
        ref = osamp.integrated_time(np.asarray(d["chain"], np.float64)[:n_last]).max()
        assert n_last == 1000 and abs(tau_max - ref) < 2e-3 * ref, (tau_max, ref, n_last)
    return d["chain"].shape, th.reshape(-1, ndim).mean(0), th.reshape(-1, ndim).std(0), acc.sum(), acc.reshape(-1, nw)[-1]


@pytest.mark.parametrize("exchange", ["", "allgather"])
def test_emcee_driver_over_two_ranks(tmp_path, monkeypatch, exchange):
    """The reference's emcee driver on two ranks, 256 walkers, 128 per rank.  Default: every rank a sub-ensemble of its own
    (partners from the LOCAL complementary half, no collective inside an iteration), chain blocks gathered once per check;
    on request (``exchange="allgather"``) ONE ensemble with the partners of both ranks.  Either way rank 0 alone writes
    chemcee_256.h5 in the layout the one-rank run writes ([iterations, 256, ndim]) and decides when to stop from the
    statistics over all 256 walkers; both ranks read the same file afterwards."""
    from test_gpu_sampling import _gaussian_33
    monkeypatch.setenv("LINNA_TEST_SHARED_DIR", str(tmp_path))
    monkeypatch.setenv("LINNA_TEST_EXCHANGE", exchange)
    ndim, means, cov, priors = _gaussian_33()
    res = _run(_driver_job)
    sig = np.sqrt(np.diag(cov))
    assert res[0][0] == res[1][0] == (1000, 256, ndim)
    np.testing.assert_array_equal(res[0][1], res[1][1])
    assert np.max(np.abs(res[0][1] - means) / sig) < 0.3 and res[0][3] > 0     # 400 steps x 256 walkers, tau ~ 100: a plumbing check
    assert np.array_equal(np.asarray(res[0][4]), np.asarray(res[1][4]))
    np.testing.assert_allclose(res[0][2], sig, rtol=0.3)
    assert sorted(os.listdir(tmp_path)) == ["chemcee_256.h5"]


def _ml_sampler_job(rank, world):
    import contextlib
    import io
    from test_gpu_callbacks import _problem2d, _core
    means, cov, priors = _problem2d(gauss=False)
    out = os.environ["LINNA_TEST_SHARED_DIR"] + "/run/"
    with contextlib.redirect_stdout(io.StringIO()) as log:
        chain, lp = _core(out, priors, means, cov, nwalkers=16, params={"nimp": 100})
    files = sorted(os.listdir(out + "iter_1"))
    return chain, np.asarray(lp), files, log.getvalue().count("rank %d of %d" % (rank, world))


def test_ml_sampler_core_under_two_ranks(tmp_path, monkeypatch):
    """``ml_sampler_core`` called on every rank of a two-rank run (what `torchrun script.py` gives): `dist.init()` brings
    the ranks up, rank 0 designs the points / calls the theory / owns every file, both ranks train (data parallel, one
    gradient all-reduce per step) and sample their shard of the 16 walkers, and both return the same chain -- the
    importance subsample rank 0 drew -- read from the one run directory."""
    monkeypatch.setenv("LINNA_TEST_SHARED_DIR", str(tmp_path))
    res = _run(_ml_sampler_job)
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(res[0][1], res[1][1])
    assert res[0][0].shape == (100, 2) and np.all(np.isfinite(res[0][0]))
    assert res[0][2] == res[1][2] and "chemcee_256.h5" in res[0][2] and "best.pth.tar" in res[0][2] and "finish.pkl" in res[0][2]
    assert res[0][3] == 1 and res[1][3] == 1                       # each rank announced its transport once
    d = np.load(str(tmp_path) + "/run/weight_im.npy")
    assert d.shape == (3, 100) and abs(d[2].sum() - 1.0) < 1e-9


def _zeus_driver_job(rank, world):
    import contextlib
    import io
    from linna_amd import sampler, util
    from test_gpu_sampling import identity_emulator_logprob, _gaussian_33
    ndim, means, cov, priors = _gaussian_33()
    lp = identity_emulator_logprob(ndim, means, cov, priors)
    out = os.environ["LINNA_TEST_SHARED_DIR"]
    nw = 136                                                          # 68 per rank (> 2 ndim in total)
    x0 = util.invTransform(priors)(means)[None, :] + 0.001 * np.random.RandomState(3 + rank).standard_normal((nw, ndim))
    drv = sampler.ZeusSampler(lp, ndim, nw, x0=x0, transform=util.Transform(priors), seed=2, exchange="allgather")
    with contextlib.redirect_stdout(io.StringIO()):
        drv.sample(None, 400, outdir=out, ntimes=1e9, tautol=1e-9)
    d = sampler.ChainStore.load(os.path.join(out, "zeus_256.h5"))
    ens = drv.sampler
    th = np.asarray(d["chain_transformed"])[200:]
    return (d["chain"].shape, th.reshape(-1, ndim).mean(0), th.reshape(-1, ndim).std(0), bool(ens._fast_ok), ens.noverflow, ens.tune,
            ens.mu, getattr(ens, "tune_off_iteration", None), ens._fast_steps)


def test_zeus_driver_shards_one_ensemble_over_the_ranks(tmp_path, monkeypatch):
    """The reference's default sampler on two ranks: one ensemble of 136 walkers, 68 per rank, slice directions from both
    ranks' complementary halves; the walkers start in a 1e-3 ball (util.py:937), so the first iterations overflow the
    one-call half step's rounds and BOTH ranks fall back to the round loop together (a decision taken over the ranks:
    collectives follow it), later iterations run the one-call path; rank 0 writes zeus_256.h5."""
    from test_gpu_sampling import _gaussian_33
    monkeypatch.setenv("LINNA_TEST_SHARED_DIR", str(tmp_path))
    ndim, means, cov, priors = _gaussian_33()
    res = _run(_zeus_driver_job)
    sig = np.sqrt(np.diag(cov))
    assert res[0][0] == res[1][0] == (400, 136, ndim)
    np.testing.assert_array_equal(res[0][1], res[1][1])
    assert np.max(np.abs(res[0][1] - means) / sig) < 0.3
    np.testing.assert_allclose(res[0][2], sig, rtol=0.3)
    assert res[0][3] and res[1][3]                                   # the one-call half step ran on both ranks
    assert res[0][4] == res[1][4]                                    # ... and they redid the same runs on the round loop
    # ONE mu for the ONE ensemble: tuned from the counts of both ranks, so both carry the same value, leave tuning at the
    # same iteration and switch to the one-call path at the same iteration (its threshold counts the whole ensemble)
    assert res[0][5] is False and res[1][5] is False
    assert res[0][6] == res[1][6] and res[0][7] == res[1][7] and res[0][7] is not None
    assert res[0][8] == res[1][8] > 0
    assert sorted(os.listdir(tmp_path)) == ["zeus_256.h5"]


def _zeus_local_job(rank, world):
    import contextlib
    import io
    from linna_amd import sampler, util
    from oracle import sampling as osamp
    from test_gpu_sampling import identity_emulator_logprob, _gaussian_33
    ndim, means, cov, priors = _gaussian_33()
    lp = identity_emulator_logprob(ndim, means, cov, priors)
    out = os.environ["LINNA_TEST_SHARED_DIR"]
    nw = int(os.environ["LINNA_TEST_NW"])
    x0 = util.invTransform(priors)(means)[None, :] + 0.001 * np.random.RandomState(3).standard_normal((nw, ndim))
    drv = sampler.ZeusSampler(lp, ndim, nw, x0=x0, transform=util.Transform(priors), seed=2)
    log = io.StringIO()
    with contextlib.redirect_stdout(log):
        store = drv.sample(None, 600, outdir=out, ntimes=1e9, tautol=1e-9)
    d = sampler.ChainStore.load(os.path.join(out, "zeus_256.h5"))
    ens = drv.sampler
    chain = np.asarray(d["chain"])
    th = np.asarray(d["chain_transformed"])[300:]
    per_rank_mean = [th[:, r * (nw // world):(r + 1) * (nw // world)].reshape(-1, ndim).mean(0) for r in range(world)]
    # the convergence statistics of the drivers see ALL walkers: tau of the stored chain as the oracle's estimator computes it
    tau = osamp.integrated_time(chain[120:]) if rank == 0 else None
    return (chain.shape, th.reshape(-1, ndim).mean(0), th.reshape(-1, ndim).std(0), per_rank_mean,
            None if ens is None else (ens.nw, ens.exchange, ens.mu, ens.tune, bool(ens._fast_ok), ens.world), tau)


def test_zeus_driver_gives_every_rank_a_sub_ensemble_by_default(tmp_path, monkeypatch):
    """The default multi-rank mode: 264 walkers = two sub-ensembles of 132 (>= 2 x 33), slice directions
    from the LOCAL complementary half, mu tuned PER SUB-ENSEMBLE (two different values), no collective inside an iteration;
    the merged chain in rank 0's zeus_256.h5 -- layout [iterations, 264, ndim], walker blocks in rank order -- carries the
    33-D posterior within the bounds of the one-ensemble test, and so does each rank's half of it."""
    from test_gpu_sampling import _gaussian_33
    monkeypatch.setenv("LINNA_TEST_SHARED_DIR", str(tmp_path))
    monkeypatch.setenv("LINNA_TEST_NW", "264")
    ndim, means, cov, priors = _gaussian_33()
    res = _run(_zeus_local_job)
    sig = np.sqrt(np.diag(cov))
    assert res[0][0] == res[1][0] == (600, 264, ndim)
    np.testing.assert_array_equal(res[0][1], res[1][1])              # both ranks read the same file
    assert np.max(np.abs(res[0][1] - means) / sig) < 0.3
    np.testing.assert_allclose(res[0][2], sig, rtol=0.3)
    for r in (0, 1):                                                 # each sub-ensemble on its own, too (half the samples)
        assert np.max(np.abs(res[0][3][r] - means) / sig) < 0.45
    e0, e1 = res[0][4], res[1][4]
    assert e0[0] == e1[0] == 132 and e0[1] == e1[1] == "none" and e0[5] == 2
    assert e0[3] is False and e1[3] is False and e0[4] and e1[4]     # both tuned, both on the one-call path
    assert e0[2] != e1[2] and 0.3 < e0[2] / e1[2] < 3.0              # a mu per sub-ensemble
    assert np.all(np.isfinite(res[0][5])) and np.all(res[0][5] > 1)  # tau over all 264 walkers
    assert sorted(os.listdir(tmp_path)) == ["zeus_256.h5"]


def test_an_ensemble_too_small_to_split_runs_on_rank_0(tmp_path, monkeypatch):
    """128 walkers on two ranks would leave 64 < 2 x 33 per rank: the whole ensemble runs on rank 0 ("root"), rank 1 waits at
    the end of the call and reads the same file."""
    from test_gpu_sampling import _gaussian_33
    monkeypatch.setenv("LINNA_TEST_SHARED_DIR", str(tmp_path))
    monkeypatch.setenv("LINNA_TEST_NW", "128")
    ndim, means, cov, priors = _gaussian_33()
    res = _run(_zeus_local_job)
    assert res[0][0] == res[1][0] == (600, 128, ndim)
    np.testing.assert_array_equal(res[0][1], res[1][1])
    assert res[0][4][0] == 128 and res[1][4] is None                 # rank 1 never built a sampler
    sig = np.sqrt(np.diag(cov))
    assert np.max(np.abs(res[0][1] - means) / sig) < 0.35


def _bench2(extra_env, *argv):
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "30", "--no-secondary",
                           "--no-cpu-baseline"] + list(argv), capture_output=True, text=True, env=e, timeout=600)


def test_two_rank_bench_line_and_its_watchdog():
    """`python bench.py --gpus 2` from a bare shell on the GPU box (two ranks share the device, gloo carries the collectives):
    ONE JSON line with the whole-job value, the strong-scaling, training and ensemble sections.  With the watchdog's budget
    set to nothing, the sections behind the headline are cut off: the line still comes out -- headline and roofline intact,
    the reason under "watchdog" (a collective that hangs on a node this code has never seen must not cost the driver its
    scaling point) -- and the launcher exits NON-ZERO: rank 0 leaves with bench._Watchdog.EXIT_HANG, so the driver's `rc`
    does not call a hung section a clean run."""
    import json
    r = _bench2({})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 1e7 and d["scaling"] == "weak" and "watchdog" not in d
    assert d["strong_scaling"]["nwalkers_per_gpu"] == 2048 and "error" not in d["training"] and "error" not in d["mcmc"]
    assert abs(d["value"] - 2 * 4096 * 30 / (d["ms_per_step"] * 1e-3 * 30)) < 1e-6 * d["value"]
    w = _bench2({"LINNA_BENCH_WATCHDOG_S": "0.5"})
    assert w.returncode != 0, "a cut-off run must not exit 0"
    assert "watchdog: leaving at stage" in w.stderr, w.stderr[-3000:]
    lines = [ln for ln in w.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    dw = json.loads(lines[0])
    assert "watchdog" in dw and dw["n_gpus"] == 2 and dw["value"] > 1e7 and dw["roofline"]["frac"] > 0.5
