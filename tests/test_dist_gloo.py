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


def test_bench_watchdog_prints_the_line_and_leaves_non_zero():
    """bench._Watchdog (N > 1): a section that hangs -> the held headline line on stdout with the reason under "watchdog",
    exit code EXIT_HANG (3); a section that raises -> the line, exit code EXIT_RAISED (4); a run that emits normally is not
    touched by the timer.  (The two-rank run of it is tests/test_gpu_dist.py::test_two_rank_bench_line_and_its_watchdog.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "d = bench._Watchdog(0.2, 0); d.arm({'metric': 'm', 'value': 1.0}); d.stage = 'training'\n"
            "mode = sys.argv[1]\n"
            "if mode == 'hang': time.sleep(30)\n"
            "if mode == 'raise':\n"
            "    try: raise RuntimeError('collective broke')\n"
            "    except Exception as e: d.failed(e)\n"
            "if mode == 'ok': d.emit({'metric': 'm', 'value': 2.0}); time.sleep(0.6); sys.exit(0)\n") % root
    out = {}
    for mode in ("hang", "raise", "ok"):
        r = subprocess.run([sys.executable, "-c", prog, mode], capture_output=True, text=True, timeout=120)
        lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
        assert len(lines) == 1, (mode, r.stdout, r.stderr[-1000:])
        out[mode] = (r.returncode, json.loads(lines[0]))
    assert out["hang"][0] == 3 and "did not finish" in out["hang"][1]["watchdog"] and "training" in out["hang"][1]["watchdog"]
    assert out["raise"][0] == 4 and "collective broke" in out["raise"][1]["watchdog"]
    assert out["ok"][0] == 0 and "watchdog" not in out["ok"][1] and out["ok"][1]["value"] == 2.0


def _slow_rank0_worker(rank, world, port, ret):
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      LINNA_PG_TIMEOUT_S="3")                 # the DATA path's timeout: a wait that long there is a hang
    from linna_amd import dist as ldist
    assert ldist.init(backend="gloo", comm=False) == world
    cg = ldist.control_group()
    assert cg is not None and dist.get_backend(cg) == "gloo"
    t0 = time.time()
    if rank == 0:
        time.sleep(8.0)                                       # rank 0 alone in the user's theory code (main.py:110)
    ldist.barrier()                                           # ranks > 0 wait here, past the data path's 3 s
    waited = time.time() - t0
    ok = ldist.agree(rank == 0)                               # rank 0's answer everywhere
    if rank == 0:
        time.sleep(5.0)                                       # ... and in the nimp theory evaluations (main.py:297-334)
    obj = ldist.broadcast_object({"chain": [1, 2, 3]} if rank == 0 else None)
    x = torch.ones(4) * (rank + 1)                            # the data path still works, on its own (short-timeout) group
    ldist.allreduce_grads(x)
    ret[rank] = (waited, ok, obj, float(x[0]))
    ldist.shutdown()


def test_control_plane_waits_do_not_sit_in_the_data_path_group():
    """ADVICE r3 (main.py:110): ranks > 0 park in a barrier while rank 0 alone runs generate_training_point / the nimp
    theory evaluations.  On the data path's group that wait is bounded by the group's timeout (NCCL: 10 minutes, then the
    watchdog of the WAITING ranks aborts the job); dist.barrier / agree / broadcast_object therefore run on a gloo side group
    with a week's timeout.  Here the data path's timeout is 3 s and rank 0 is 8 s and 5 s late: nothing times out."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_slow_rank0_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret[1][0] >= 7.0                                   # rank 1 really waited past the 3 s
    for r in (0, 1):
        assert ret[r][1] is True and ret[r][2] == {"chain": [1, 2, 3]} and ret[r][3] == 3.0


def _ranks_job(rank, world):
    from linna_amd import sampler
    out = {}
    for nw, ndim, ex in [(256, 33, None), (128, 33, None), (66, 33, None), (4, 2, None), (6, 2, None), (256, 33, "allgather"),
                         (256, 33, "root"), (128, 33, "local")]:
        rk = sampler._Ranks(nw, None, ndim, ex)
        out[(nw, ndim, ex)] = (rk.mode, rk.active, rk.world, rk.nw, rk.exchange, len(rk.mine(np.zeros((nw, ndim)))))
    try:
        sampler._Ranks(130, None, 3, "allgather")
        out["odd"] = "accepted"
    except ValueError as e:
        out["odd"] = str(e)
    return out


def test_how_a_driver_splits_its_ensemble_over_two_ranks():
    """sampler._Ranks: per-rank sub-ensembles with the LOCAL complementary half whenever a rank's share is
    a valid ensemble of its own (>= 2 ndim walkers), the whole ensemble on rank 0 below that, the per-half-step all-gather
    only on request."""
    r0, r1 = _run(_ranks_job)
    for r, rank in ((r0, 0), (r1, 1)):
        assert r[(256, 33, None)] == ("local", True, 2, 128, "none", 128)
        assert r[(128, 33, None)][0] == "root"                       # 64 walkers per rank < 2 x 33
        assert r[(66, 33, None)][0] == "root"                        # (66 is not a multiple of 4 either)
        assert r[(4, 2, None)] == ("root", rank == 0, 1, 4, "none", 4)   # the reference's own test size: rank 0 alone
        assert r[(6, 2, None)][0] == "root"
        assert r[(256, 33, "allgather")] == ("allgather", True, 2, 128, "allgather", 128)
        assert r[(256, 33, "root")] == ("root", rank == 0, 1, 256, "none", 256)
        assert r[(128, 33, "local")] == ("local", True, 2, 64, "none", 64)   # forced below the floor: the caller's business
        assert "cannot be split" in r["odd"]


def _enter_worker(rank, world, port, ret):
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from linna_amd import dist as ldist
    assert ldist.init(backend="gloo", comm=False) == world
    ldist.enter("both ranks arrive", timeout=20.0)                # every rank arrives: returns
    t0 = time.time()
    if rank == 0:
        try:
            ldist.enter("sampler.ZeusSampler.sample", timeout=3.0)     # rank 1 never makes this call
            ret[rank] = ("returned", time.time() - t0)
        except RuntimeError as e:
            ret[rank] = (str(e), time.time() - t0)
    else:
        time.sleep(6.0)
        ret[rank] = ("absent", time.time() - t0)
    os._exit(0)                                                   # (the side group is broken by design after the timeout: no shutdown)


def test_a_driver_called_on_a_subset_of_ranks_raises_within_seconds():
    """ZeusSampler.sample called on rank 0 only used to park on the control plane's gloo group,
    whose timeout is a week.  dist.enter (first line of both drivers) bounds the wait and names the call."""
    mgr = mp.Manager()
    ret = mgr.dict()
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_enter_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    msg, dt = ret[0]
    assert "sampler.ZeusSampler.sample was entered by rank 0, but not by every one of the 2 ranks within 3 s" in msg, msg
    assert "must be made on every rank" in msg and 2.5 < dt < 15.0
    assert ret[1][0] == "absent"
