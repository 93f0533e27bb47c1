"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" on CPU for tests).  The hot path shards without data-path collectives:

* walkers: every rank owns ``nwalkers`` walkers and advances them independently; chain state
  is gathered to all ranks once per flush (``gather_chain``); optionally the complementary
  half-ensemble is all-gathered per half step (``EnsembleSampler(exchange="allgather")``).
* training: every rank takes its own batch of B rows per step (``rank_batches``), gradients
  are summed with ONE all-reduce over the flat fp32 gradient buffer (``allreduce_grads``); the
  per-rank loss gradient is already scaled by 1/(B * world) so the sum is the global mean, and
  the learning rate follows the reference's ``lr * size`` rule (predictor_gpu.py:246).
"""
import os

import torch
import torch.distributed as dist


def init(backend=None, device=None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / MASTER_*); no-op for 1 rank."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 or dist.is_initialized():
        return world
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
    dist.init_process_group(backend, **kw)
    return world


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def rank(group=None):
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def rank_batches(batches, rank, size):
    """Step s of rank r uses global batch s*size + r: disjoint batches, ``len(batches)//size`` steps."""
    nsteps = len(batches) // size
    return [batches[s * size + rank] for s in range(nsteps)]


def allreduce_grads(flat_grad, loss_scalar=None, group=None):
    """Sum the flat gradient buffer (and the scalar loss) over ranks, in place."""
    if world_size(group) == 1:
        return
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    if loss_scalar is not None:
        dist.all_reduce(loss_scalar, op=dist.ReduceOp.SUM, group=group)


def gather_chain(chain, lps, group=None):
    """``chain[n, nw, ndim]``, ``lps[n, nw]`` of every rank -> ``[n, world*nw, ndim]``, ``[n, world*nw]``
    on every rank (walker blocks ordered by rank)."""
    w = world_size(group)
    if w == 1:
        return chain, lps
    n, nw, nd = chain.shape
    allc = torch.empty((w * n, nw, nd), dtype=chain.dtype, device=chain.device)     # concatenation along dim 0
    alll = torch.empty((w * n, nw), dtype=lps.dtype, device=lps.device)
    dist.all_gather_into_tensor(allc, chain.contiguous(), group=group)
    dist.all_gather_into_tensor(alll, lps.contiguous(), group=group)
    return (allc.view(w, n, nw, nd).permute(1, 0, 2, 3).reshape(n, w * nw, nd),
            alll.view(w, n, nw).permute(1, 0, 2).reshape(n, w * nw))


def gather_rows(x, group=None):
    """All-gather of ``x[m, ...]`` along dim 0 (complementary walkers of every rank)."""
    w = world_size(group)
    if w == 1:
        return x
    out = torch.empty((w * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x.contiguous(), group=group)
    return out
