"""Two ranks on the ONE GPU of the test box (gloo; RCCL refuses two ranks on one device): the multi-rank
paths of the sampler and of the trainer on device tensors -- gradient all-reduce, complementary-ensemble
all-gather, chain gather -- against the single-rank results they must reproduce."""
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
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


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
    np.testing.assert_allclose(res[0][0], ref, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(res[0][1], float(eng.loss_mean.item()), rtol=1e-4)


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
