"""Checkpoint I/O in the reference's on-disk format (linna/nnutils.py:109-151).

``last.pth.tar`` / ``best.pth.tar`` are ``torch.save`` files holding
``{'epoch', 'state_dict', 'optim_dict'}`` with the reference's state_dict key names, so
checkpoints written by either implementation load in the other.
"""
import os
import shutil

import torch


def save_checkpoint(state, is_best, checkpoint):
    """nnutils.py:109-126."""
    if not os.path.exists(checkpoint):
        os.makedirs(checkpoint)
    filepath = os.path.join(checkpoint, "last.pth.tar")
    torch.save(state, filepath)
    if is_best:
        shutil.copyfile(filepath, os.path.join(checkpoint, "best.pth.tar"))


def save_state(state, filepath):
    """One checkpoint file (the layout of nnutils.py:109-126: {epoch, state_dict, optim_dict})."""
    d = os.path.dirname(filepath)
    if d and not os.path.exists(d):
        os.makedirs(d)
    torch.save(state, filepath)


def _numpy_scalar_globals():
    """The reconstructors a ``torch.optim`` state dict needs when ``lr`` came from ``np.load(lr.npy)``
    (SURVEY §8 a19): numpy scalars / arrays of the plain float / int dtypes (data-only constructors), nothing else."""
    import numpy as np
    core = getattr(np, "_core", None) or np.core
    out = [core.multiarray.scalar, core.multiarray._reconstruct, np.ndarray, np.dtype]
    for name in ("Float64DType", "Float32DType", "Int64DType", "Int32DType", "BoolDType"):
        t = getattr(getattr(np, "dtypes", None), name, None)
        if t is not None:
            out.append(t)
    return out


def read_checkpoint(path, device=None):
    """A checkpoint file is only ever read with ``weights_only=True`` (tensors, containers, plain numbers --
    nothing in the file is executed); files whose ``optim_dict`` holds numpy scalars get exactly those
    reconstructors allowed, and a file that still does not load is refused."""
    if not os.path.exists(path):
        raise FileNotFoundError("File doesn't exist {}".format(path))
    where = device if device is not None else "cpu"
    try:
        return torch.load(path, map_location=where, weights_only=True)
    except Exception:
        with torch.serialization.safe_globals(_numpy_scalar_globals()):
            return torch.load(path, map_location=where, weights_only=True)


def load_checkpoint(checkpoint, model, optimizer=None, device=None, ismpi=False):
    """nnutils.py:129-151: restore ``model`` (and ``optimizer`` when given) from a file."""
    ck = read_checkpoint(checkpoint, "cpu")
    model.load_state_dict(ck["mpi_state_dict"] if ismpi else ck["state_dict"])
    if optimizer is not None and hasattr(optimizer, "load_state_dict") and "optim_dict" in ck:
        optimizer.load_state_dict(ck["optim_dict"])
    return ck
