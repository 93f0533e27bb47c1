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


def read_checkpoint(path, device=None):
    if not os.path.exists(path):
        raise FileNotFoundError("File doesn't exist {}".format(path))
    # optim_dict of reference-written files can hold numpy scalars: full unpickle
    return torch.load(path, map_location=device if device is not None else "cpu", weights_only=False)


def load_checkpoint(checkpoint, model, optimizer=None, device=None, ismpi=False):
    """nnutils.py:129-151: restore ``model`` (and ``optimizer`` when given) from a file."""
    ck = read_checkpoint(checkpoint, "cpu")
    model.load_state_dict(ck["mpi_state_dict"] if ismpi else ck["state_dict"])
    if optimizer is not None and hasattr(optimizer, "load_state_dict") and "optim_dict" in ck:
        optimizer.load_state_dict(ck["optim_dict"])
    return ck
