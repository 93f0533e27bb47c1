"""CPU, world_size 2, gloo: the multi-rank conventions (batch sharding + gradient all-reduce
reproduce the single-rank global-batch gradient; chain gather layout).  Compute inside the
ranks is the numpy oracle -- only the distributed plumbing of linna_amd.dist is under test."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
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


def _grad_job(rank, world):
    from oracle import emulator, training
    from linna_amd import dist as ldist
    p = cases.training_problem("train_mlp_7_5")
    g = cases.golden("train_mlp_7_5")
    B = p["X"].shape[1]
    X, Y = p["X"].reshape(3 * B, -1), p["Y"].reshape(3 * B, -1)
    # 6 batches of B/2 rows; rank r takes batches s*world + r
    hb = B // 2
    batches = [torch.arange(i * hb, (i + 1) * hb) for i in range(6)]
    mine = ldist.rank_batches(batches, rank, world)
    assert len(mine) == 3
    rows = mine[0].numpy()
    x = (X[rows] - p["X_mean"][None, :]) / p["X_std"][None, :]
    pred, caches = emulator.forward(p["weights"], x, p["kind"], p["nin"], p["nout"], keep=True, **p["kw"])
    _, dpred = training.loss_grad(pred, Y[rows], g["data_norm"].reshape(-1), g["icov_norm"], p["sigma"].astype(np.float32),
                                  p["y_mean"], p["y_std"])
    dpred = dpred / world          # the engine's inv_batch = 1/(B_local * world)
    _, grads = emulator.backward(p["weights"], caches, dpred.astype(np.float32), p["kind"], p["nin"], p["nout"], **p["kw"])
    flat = torch.from_numpy(np.concatenate([grads[k].ravel() for k in sorted(grads)]).astype(np.float32))
    ldist.allreduce_grads(flat, None)
    return flat.numpy()


def test_sharded_gradient_equals_global_batch_gradient():
    from oracle import emulator, training
    out = _run(_grad_job)
    np.testing.assert_array_equal(out[0], out[1])
    p = cases.training_problem("train_mlp_7_5")
    g = cases.golden("train_mlp_7_5")
    B = p["X"].shape[1]
    x = (p["X"][0] - p["X_mean"][None, :]) / p["X_std"][None, :]      # rows 0..B-1 = the two ranks' first batches
    pred, caches = emulator.forward(p["weights"], x, p["kind"], p["nin"], p["nout"], keep=True, **p["kw"])
    _, dpred = training.loss_grad(pred, p["Y"][0], g["data_norm"].reshape(-1), g["icov_norm"], p["sigma"].astype(np.float32),
                                  p["y_mean"], p["y_std"])
    _, grads = emulator.backward(p["weights"], caches, dpred, p["kind"], p["nin"], p["nout"], **p["kw"])
    ref = np.concatenate([grads[k].ravel() for k in sorted(grads)])
    np.testing.assert_allclose(out[0], ref, rtol=2e-4, atol=1e-6 * np.abs(ref).max())


def _chain_job(rank, world):
    from linna_amd import dist as ldist
    n, nw, nd = 5, 4, 3
    chain = torch.full((n, nw, nd), float(rank)) + torch.arange(nw, dtype=torch.float32)[None, :, None] * 0.1
    lps = torch.full((n, nw), float(rank))
    c, l = ldist.gather_chain(chain, lps)
    comp = ldist.gather_rows(torch.full((2, 4), float(rank)))
    return c.numpy(), l.numpy(), comp.numpy()


def test_chain_gather_layout():
    out = _run(_chain_job)
    for c, l, comp in out:
        assert c.shape == (5, 8, 3) and l.shape == (5, 8)
        assert np.all(l[:, :4] == 0) and np.all(l[:, 4:] == 1)       # walker blocks ordered by rank
        np.testing.assert_allclose(c[2, 5, 0], 1.1)
        assert comp.shape == (4, 4) and np.all(comp[:2] == 0) and np.all(comp[2:] == 1)
    np.testing.assert_array_equal(out[0][0], out[1][0])


def test_single_rank_helpers_are_noops():
    from linna_amd import dist as ldist
    assert ldist.world_size() == 1 and ldist.rank() == 0
    x = torch.ones(3)
    ldist.allreduce_grads(x)
    assert torch.equal(x, torch.ones(3))
    c, l = ldist.gather_chain(torch.zeros(2, 2, 2), torch.zeros(2, 2))
    assert c.shape == (2, 2, 2)


def _bench(*argv, **env):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), capture_output=True, text=True, env=e, timeout=300)


def test_bench_launches_its_own_ranks_from_a_bare_shell():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the driver's N > 1 form): bench.py starts the two
    ranks itself as child processes, they rendezvous through linna_amd.dist.init() and rank 0's JSON line comes back on
    stdout -- alone.  A rank that dies makes the launcher exit non-zero."""
    import json
    r = _bench("--gpus", "2", "--backend", "gloo", "--launch-check")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["launch_check"] and d["world"] == 2 and d["rank_sum"] == 3.0 and "gloo" in d["collectives"]
    bad = _bench("--gpus", "2", "--backend", "gloo", "--launch-check", LINNA_BENCH_FAIL_RANK="1")
    assert bad.returncode != 0 and not bad.stdout.strip()
