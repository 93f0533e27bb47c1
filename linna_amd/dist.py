"""Multi-GPU plumbing: one process per GPU.  The DATA PATH collectives -- gradient all-reduce, walker / chain
all-gather -- go through the library's own RCCL communicator (``linna_comm_init`` / ``linna_allreduce_sum_f32`` /
``linna_allgather_f32`` of include/linna_hip.h, enqueued on the caller's stream) once ``comm_init()`` has been called;
``torch.distributed`` is the control plane (rendezvous, hand-off of the RCCL unique id, host-side barriers) and the
fallback transport where RCCL cannot run: CPU tensors and the gloo tests, including two ranks sharing the single GPU of
a test box (RCCL refuses two ranks on one device).  The hot path shards without data-path collectives:

* walkers: every rank owns ``nwalkers`` walkers and advances them independently; chain state
  is gathered to all ranks once per flush (``gather_chain``); optionally the complementary
  half-ensemble is all-gathered per half step (``EnsembleSampler(exchange="allgather")``).
* training: every rank takes its own batch of B rows per step (``rank_batches``), gradients
  are summed with ONE all-reduce over the flat fp32 gradient buffer (``allreduce_grads``); the
  per-rank loss gradient is already scaled by 1/(B * world) so the sum is the global mean, and
  the learning rate follows the reference's ``lr * size`` rule (predictor_gpu.py:246).
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib

_comm = {}          # device index -> (rank, world) of the RCCL communicator held by that device's context


def comm_init(device_index=None, group=None, rank=None, world=None, unique_id=None, timeout=None):
    """Create this rank's RCCL communicator (collective over the ranks of ``group``).  Rank 0 draws the unique id and
    every rank receives it through torch.distributed's object broadcast (any backend), unless ``unique_id`` bytes are
    handed in (a file, a store).  One rank per device: RCCL refuses two ranks on one GPU."""
    if device_index is None:
        device_index = torch.cuda.current_device()
    if device_index in _comm:
        return _comm[device_index]
    if world is None:
        world, rank = world_size(group), globals()["rank"](group)
    if unique_id is None:
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        if rank == 0:
            _lib.call("linna_comm_unique_id", buf)
        box = [buf.raw]
        if world > 1:
            dist.broadcast_object_list(box, src=0 if group is None else dist.get_global_rank(group, 0), group=group)
        unique_id = box[0]
    if len(unique_id) != _lib.COMM_ID_BYTES:
        raise ValueError("an RCCL unique id is %d bytes" % _lib.COMM_ID_BYTES)
    ctx = _lib.ctx(device_index)
    idbuf = C.create_string_buffer(unique_id, _lib.COMM_ID_BYTES)

    def init():
        with torch.cuda.device(device_index):
            _lib.call("linna_comm_init", ctx, int(rank), int(world), idbuf)
    if timeout is None:
        init()
    else:
        # ncclCommInitRank blocks until every rank has arrived; a rank that never does would hang the job.  Run it on a
        # thread and give up after `timeout` seconds (the caller then keeps torch.distributed as the transport).
        import threading
        err = []

        def run():
            try:
                init()
            except Exception as e:                          # noqa: BLE001
                err.append(e)
        t = threading.Thread(target=run, name="linna-comm-init", daemon=True)
        t.start()
        t.join(timeout)
        if t.is_alive():
            raise _lib.LinnaHipError("linna_comm_init did not return within %.0f s" % timeout)
        if err:
            raise err[0]
    _comm[device_index] = (int(rank), int(world))
    return _comm[device_index]


def comm_active(t):
    """True when ``t`` is a device tensor whose device holds an RCCL communicator."""
    return torch.is_tensor(t) and t.is_cuda and t.device.index in _comm


def comm_info(device_index=None):
    """(rank, nranks, rccl version) as the library reports them; nranks 0 = no communicator."""
    if device_index is None:
        device_index = torch.cuda.current_device()
    r, n, v = C.c_int(), C.c_int(), C.c_int()
    _lib.call("linna_comm_info", _lib.ctx(device_index), C.byref(r), C.byref(n), C.byref(v))
    return r.value, n.value, v.value


def comm_destroy(device_index=None):
    for d in ([device_index] if device_index is not None else list(_comm)):
        if d in _comm:
            _lib.call("linna_comm_destroy", _lib.ctx(d))
            del _comm[d]


def comm_forget(device_index=None):
    """Stop routing collectives through a communicator WITHOUT destroying it (one that timed out may still be blocked in
    RCCL: tearing it down could block as well)."""
    for d in ([device_index] if device_index is not None else list(_comm)):
        _comm.pop(d, None)


def comm_selftest(device_index=None, timeout=60.0):
    """One ``linna_allreduce_sum_f32`` of 1024 floats and one ``linna_allgather_f32`` on a side stream, waited for at most ``timeout`` seconds and
    checked against the closed form: True when this rank's communicator works.  (A launcher calls this before it
    commits the data path to the communicator; every rank must then agree, e.g. by a MIN all-reduce of the answers.)"""
    import time
    if device_index is None:
        device_index = torch.cuda.current_device()
    if device_index not in _comm:
        return False
    r, w = _comm[device_index]
    dev = torch.device("cuda", device_index)
    side = torch.cuda.Stream(device=dev)
    done = torch.cuda.Event()
    with torch.cuda.stream(side):
        t = torch.full((1024,), float(r + 1), dtype=torch.float32, device=dev)
        _lib.call("linna_allreduce_sum_f32", _lib.ctx(device_index), _f32(t), t.numel(), C.c_void_p(side.cuda_stream))
        # ... and one all-gather (the walker exchange of a half step, the chain gather of a flush)
        g_in = torch.full((256,), float(r + 1), dtype=torch.float32, device=dev)
        g_out = torch.zeros((w * 256,), dtype=torch.float32, device=dev)
        _lib.call("linna_allgather_f32", _lib.ctx(device_index), _f32(g_in), _f32(g_out), g_in.numel(), C.c_void_p(side.cuda_stream))
        done.record(side)
    t0 = time.perf_counter()
    while not done.query():
        if time.perf_counter() - t0 > timeout:
            return False
        time.sleep(0.005)
    want = torch.arange(1, w + 1, dtype=torch.float32, device=dev).repeat_interleave(256)
    return bool((t == float(w * (w + 1) // 2)).all().item()) and bool((g_out == want).all().item())


def _f32(t):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise _lib.LinnaHipError("RCCL entries take contiguous float32 device tensors")
    return C.c_void_p(t.data_ptr())


_state = {"collectives": None, "control": None}

# Control plane vs data path.  The waits of the host-only phases -- ranks > 0 parked while rank 0 alone designs training
# points and runs the user's theory code (main.py:110, 297-334: minutes to hours), decisions read from the file system --
# must not sit in the data path's process group: an NCCL (RCCL) collective that waits longer than the group's timeout
# (torch 2.10: 10 minutes) makes the watchdog of the WAITING ranks abort the whole job.  `barrier`, `agree` and
# `broadcast_object` therefore run on a gloo side group with its own, long timeout (LINNA_CONTROL_TIMEOUT_S, default one
# week); the data path (gradient all-reduce, walker exchange, chain gather) keeps the short one, where a wait that long IS
# a hang.
CONTROL_TIMEOUT_S = float(os.environ.get("LINNA_CONTROL_TIMEOUT_S", str(7 * 24 * 3600)))


def control_group():
    """The gloo side group of the control plane (None for one rank / before ``init()``)."""
    return _state["control"]


def _make_control_group():
    """Create the gloo side group -- and keep it only if EVERY rank has one: ``new_group`` can fail on one node alone
    (interface selection, say); ranks that disagreed about the group would run `barrier` / `agree` / `broadcast_object` on
    different groups and wait for each other for CONTROL_TIMEOUT_S.  The answers are MIN-all-reduced on the default
    group, as `init()` does for the RCCL communicator."""
    import datetime
    import sys
    if _state["control"] is None and dist.is_initialized() and dist.get_world_size() > 1:
        grp, err = None, None
        try:
            grp = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=CONTROL_TIMEOUT_S))
        except Exception as e:                                  # noqa: BLE001
            err = e
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        have = torch.tensor([1.0 if grp is not None else 0.0], dtype=torch.float32, device=dev)
        dist.all_reduce(have, op=dist.ReduceOp.MIN)
        if float(have.item()) < 1.0:
            if grp is not None:
                sys.stderr.write("linna_amd.dist: another rank has no gloo control group: dropped here as well; control-plane "
                                 "waits stay on the %s group\n" % dist.get_backend())
            else:
                sys.stderr.write("linna_amd.dist: no gloo control group (%r); control-plane waits stay on the %s group\n" % (err, dist.get_backend()))
            grp = None
        _state["control"] = grp
    return _state["control"]


def collectives():
    """What carries the data-path collectives of this process (a sentence for logs and the bench line); None = single rank
    or ``init()`` not called."""
    return _state["collectives"]


def init(backend=None, device=None, comm=None, timeout=120.0):
    """Bring a multi-rank run up from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); no-op for one
    rank.  What `hvd.init()` / the DDP wrap would be in the reference (predictor_gpu.py:240-252, 265-266):

    1. rendezvous through ``torch.distributed`` (``backend``: nccl when this rank has a GPU of its own, else gloo);
    2. the library's own RCCL communicator through the C ABI (``linna_comm_init``, the unique id handed over the
       process group), bounded by ``timeout``;
    3. one self-test all-reduce on a side stream (``comm_selftest``), bounded as well;
    4. a MIN all-reduce of the answers: the data path runs on RCCL through the C ABI only if it came up and answered on
       EVERY rank; otherwise every rank falls back to ``torch.distributed`` as the transport (a communicator that may
       be stuck is forgotten, never torn down).

    ``comm``: True / False force or skip steps 2-4; None = attempt them when the backend is nccl (one rank per device;
    a gloo rehearsal that shares one GPU between ranks cannot form an RCCL communicator).  Returns the world size;
    ``collectives()`` tells which transport was agreed."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return world
    local_rank = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    ndev = torch.cuda.device_count()
    import datetime
    pg_kw = {}
    if os.environ.get("LINNA_PG_TIMEOUT_S"):              # the data path's timeout (default: torch's)
        pg_kw["timeout"] = datetime.timedelta(seconds=float(os.environ["LINNA_PG_TIMEOUT_S"]))
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if (ndev >= 1 and local_rank < ndev and int(os.environ.get("LOCAL_WORLD_SIZE", world)) <= ndev) else "gloo"
        if backend == "nccl":
            if local_rank >= ndev:
                raise _lib.LinnaHipError("rank %d has no GPU (%d visible): one rank per GPU" % (local_rank, ndev))
            if device is None:
                device = torch.device("cuda", local_rank)
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", device_id=torch.device(device), **pg_kw)
        else:
            if ndev:
                torch.cuda.set_device(local_rank % ndev if device is None else device)
            dist.init_process_group(backend, **pg_kw)
    _make_control_group()                                  # (collective: every rank passes here)
    backend = dist.get_backend()
    if _state["collectives"] is not None:
        return world
    if comm is None:
        comm = backend == "nccl" and os.environ.get("LINNA_COMM", "rccl") != "torch"
    if not comm or not torch.cuda.is_available():
        _state["collectives"] = "torch.distributed %s%s" % (backend, " (rehearsal: ranks may share a GPU)" if backend != "nccl" else "")
        return world
    dev_index = torch.cuda.current_device() if device is None else torch.device(device).index
    why = ""
    try:
        comm_init(dev_index, timeout=timeout)
        ok = comm_selftest(dev_index, timeout=min(60.0, timeout))
        why = "" if ok else "self-test all-reduce wrong or late"
    except Exception as e:                                          # noqa: BLE001
        ok, why = False, repr(e)[:200]
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=torch.device("cuda", dev_index) if backend == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        r, n, v = comm_info(dev_index)
        _state["collectives"] = ("RCCL %s, %d ranks, through the C ABI (linna_comm_init / linna_allreduce_sum_f32 / "
                                 "linna_allgather_f32 / linna_broadcast_f32)" % (v, n))
    else:
        if ok:
            comm_destroy(dev_index)
        else:
            comm_forget(dev_index)                                  # (never tear down a communicator that may be stuck)
        _state["collectives"] = "torch.distributed %s (the library's communicator did not come up on every rank%s)" % (
            backend, ": " + why if why else "")
    return world


def shutdown():
    """Tear down what ``init()`` brought up (communicator first, then the process group)."""
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    comm_destroy()
    _state["collectives"] = None
    _state["control"] = None
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def world_size(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group)
    return max([w for _, w in _comm.values()] + [1])


def rank(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group)
    return max([r for r, _ in _comm.values()] + [0])


def barrier(group=None):
    """Host-side barrier of the control plane (on the gloo side group when ``init()`` made one: a rank may wait here for
    hours while rank 0 runs host-only work); no-op for one rank."""
    if world_size(group) > 1 and dist.is_available() and dist.is_initialized():
        if group is None and _state["control"] is not None:
            group = _state["control"]
        dist.barrier(group=group)


ENTRY_TIMEOUT_S = float(os.environ.get("LINNA_ENTRY_TIMEOUT_S", "120"))


def enter(what, group=None, timeout=None):
    """Fail fast when a collective entry point (a sampling driver, a data-parallel training run) is called on a SUBSET of the
    ranks: every rank of ``group`` must arrive here within ``timeout`` seconds (``LINNA_ENTRY_TIMEOUT_S``, default 120), else
    the ranks that did arrive raise ``RuntimeError`` naming the call and -- on rank 0 -- the ranks that are missing.  Without
    it such a call parks on the control plane's side group, whose timeout is a week by design (rank 0 may run the user's
    theory code for hours while the others wait), silently.  ``torch.distributed.monitored_barrier`` on the gloo side
    group; a run whose control plane has no gloo group (``init()`` could not make one) is not checked."""
    if world_size(group) == 1 or not (dist.is_available() and dist.is_initialized()):
        return
    g = group
    if g is None:
        g = _state["control"] if _state["control"] is not None else (None if dist.get_backend() == "gloo" else False)
    elif dist.get_backend(g) != "gloo":
        g = False
    if g is False:
        return
    import datetime
    t = ENTRY_TIMEOUT_S if timeout is None else float(timeout)
    try:
        dist.monitored_barrier(group=g, timeout=datetime.timedelta(seconds=t), wait_all_ranks=True)
    except Exception as e:                                      # noqa: BLE001  (gloo raises RuntimeError / DistBackendError)
        raise RuntimeError("%s was entered by rank %d, but not by every one of the %d ranks within %.0f s: the call is collective "
                           "(walkers / batches are sharded over the ranks, chain blocks gathered) and must be made on every rank "
                           "of the run -- guard rank-0-only code with `if rank == 0`, not this call.  [%s]"
                           % (what, rank(group), world_size(group), t, str(e).splitlines()[0][:300])) from None


def broadcast_object(obj, src=0):
    """Rank ``src``'s picklable object on every rank, over the control plane (the result of a host-only phase)."""
    if world_size() == 1 or not (dist.is_available() and dist.is_initialized()):
        return obj
    box = [obj if rank() == src else None]
    dist.broadcast_object_list(box, src=src, group=_state["control"])
    return box[0]


def agree(flag, group=None):
    """Rank 0's boolean on every rank (decisions read from the file system are taken once, by the rank that owns the
    files, so that no rank can see a different answer and leave the others in a collective)."""
    if world_size(group) == 1 or not (dist.is_available() and dist.is_initialized()):
        return bool(flag)
    if group is None and _state["control"] is not None:
        group = _state["control"]
    box = [bool(flag)]
    dist.broadcast_object_list(box, src=0 if group is None else dist.get_global_rank(group, 0), group=group)
    return bool(box[0])


def any_rank(flag, group=None):
    """True on every rank when ``flag`` is true on at least one (a decision every rank must take the same way because
    collectives follow it)."""
    if world_size(group) == 1 or not (dist.is_available() and dist.is_initialized()):
        return bool(flag)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(int(t.item()))


def rank_batches(batches, rank, size):
    """Step s of rank r uses global batch s*size + r: disjoint batches, ``len(batches)//size`` steps."""
    nsteps = len(batches) // size
    return [batches[s * size + rank] for s in range(nsteps)]


def allreduce_grads(flat_grad, loss_scalar=None, group=None):
    """Sum the flat gradient buffer (and the scalar loss) over ranks, in place.  When the scalar sits right behind the
    gradient buffer in memory (``nn._Emulator.grad_tail``) the two travel in ONE all-reduce."""
    if comm_active(flat_grad):
        ctx, st, n = _lib.ctx(flat_grad.device.index), _lib.stream(), flat_grad.numel()
        if loss_scalar is not None and loss_scalar.data_ptr() == flat_grad.data_ptr() + 4 * n:
            _lib.call("linna_allreduce_sum_f32", ctx, _f32(flat_grad), n + 1, st)
            return
        _lib.call("linna_allreduce_sum_f32", ctx, _f32(flat_grad), n, st)
        if loss_scalar is not None:
            _lib.call("linna_allreduce_sum_f32", ctx, _f32(loss_scalar), loss_scalar.numel(), st)
        return
    if world_size(group) == 1:
        return
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    if loss_scalar is not None:
        dist.all_reduce(loss_scalar, op=dist.ReduceOp.SUM, group=group)


def _allgather(x, group):
    """``x`` of every rank concatenated along dim 0 (rank order)."""
    x = x.contiguous()
    if comm_active(x) and x.dtype == torch.float32:
        w = _comm[x.device.index][1]
        out = torch.empty((w * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        _lib.call("linna_allgather_f32", _lib.ctx(x.device.index), _f32(x), _f32(out), x.numel(), _lib.stream())
        return out, w
    w = world_size(group)
    out = torch.empty((w * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x, group=group)
    return out, w


def gather_chain(chain, lps, group=None):
    """``chain[n, nw, ndim]``, ``lps[n, nw]`` of every rank -> ``[n, world*nw, ndim]``, ``[n, world*nw]``
    on every rank (walker blocks ordered by rank)."""
    if world_size(group) == 1:
        return chain, lps
    n, nw, nd = chain.shape
    allc, w = _allgather(chain, group)
    alll, _ = _allgather(lps, group)
    return (allc.view(w, n, nw, nd).permute(1, 0, 2, 3).reshape(n, w * nw, nd),
            alll.view(w, n, nw).permute(1, 0, 2).reshape(n, w * nw))


def gather_rows(x, group=None):
    """All-gather of ``x[m, ...]`` along dim 0 (complementary walkers of every rank)."""
    if world_size(group) == 1:
        return x
    return _allgather(x, group)[0]


def broadcast_value(v, group=None, device=None):
    """A host float of rank 0 to every rank (the range-tested learning rate, predictor_gpu.py:223-245)."""
    if world_size(group) == 1:
        return float(v)
    if device is not None and torch.device(device).index in _comm:
        t = torch.tensor([float(v)], dtype=torch.float32, device=device)
        _lib.call("linna_broadcast_f32", _lib.ctx(t.device.index), _f32(t), 1, 0, _lib.stream())
        return float(t.item())
    box = [float(v)]
    dist.broadcast_object_list(box, src=0 if group is None else dist.get_global_rank(group, 0), group=group)
    return float(box[0])
