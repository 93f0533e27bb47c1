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


def _numpy_module_names():
    """(name in files of the OTHER numpy generation, name this numpy resolves): numpy 2 moved ``numpy.core`` to
    ``numpy._core``; a checkpoint written under numpy 1.x names ``numpy.core.multiarray.scalar`` and torch matches
    allowed globals by that text."""
    import numpy as np
    return (b"numpy.core.", b"numpy._core.") if getattr(np, "_core", None) is not None else (b"numpy._core.", b"numpy.core.")


def _rename_numpy_globals(pkl):
    """One pickle stream with the module text of its numpy globals renamed (GLOBAL opcodes and the string pushes a
    STACK_GLOBAL consumes); everything else byte for byte.  Nothing is unpickled: ``pickletools.genops`` only walks the
    opcodes.  Renaming changes the length of a string, so the FRAME opcodes of a protocol >= 4 stream (which promise the
    byte length of what follows) are DROPPED from the copy -- frames are optional, every unpickler reads a frameless
    stream -- instead of left stale ("pickle exhausted before end of frame").  Returns (new bytes, number of bytes of
    ``pkl`` the stream occupied)."""
    import io
    import pickletools
    import struct
    old, new = _numpy_module_names()
    ops = list(pickletools.genops(io.BytesIO(pkl)))
    end = ops[-1][2] + 1                                    # STOP is one byte
    out, last = [], 0
    for k, (op, arg, pos) in enumerate(ops):
        nxt = ops[k + 1][2] if k + 1 < len(ops) else end
        if op.name == "GLOBAL" and arg.encode().startswith(old):
            mod, name = arg.split(" ", 1)
            repl = b"c" + new + mod.encode()[len(old):] + b"\n" + name.encode() + b"\n"
        elif op.name in ("SHORT_BINUNICODE", "BINUNICODE", "BINUNICODE8") and isinstance(arg, str) and arg.encode().startswith(old):
            txt = new + arg.encode()[len(old):]
            if op.name == "BINUNICODE8":
                repl = b"\x8d" + struct.pack("<Q", len(txt))
            else:
                repl = (b"\x8c" + bytes([len(txt)])) if (op.name == "SHORT_BINUNICODE" and len(txt) < 256) else (b"X" + struct.pack("<I", len(txt)))
            repl += txt
        elif op.name == "FRAME":
            repl = b""
        else:
            continue
        out.append(pkl[last:pos]); out.append(repl)
        last = nxt
    out.append(pkl[last:end])
    return b"".join(out), end


def _renamed_copy(path):
    """The checkpoint file as an in-memory copy whose pickled numpy globals carry this numpy's module names: the
    ``data.pkl`` record of a zip checkpoint, or the leading pickles of a legacy stream."""
    import io
    import zipfile
    if zipfile.is_zipfile(path):
        buf = io.BytesIO()
        with zipfile.ZipFile(path) as zin, zipfile.ZipFile(buf, "w", zipfile.ZIP_STORED) as zout:
            for info in zin.infolist():
                raw = zin.read(info.filename)
                if info.filename.endswith("/data.pkl") or info.filename == "data.pkl":
                    raw = _rename_numpy_globals(raw)[0]
                zout.writestr(info.filename, raw)
        buf.seek(0)
        return buf
    with open(path, "rb") as f:
        raw = f.read()
    out, pos = [], 0
    for _ in range(5):                  # magic number, protocol version, sys info, the object, the storage keys
        if pos >= len(raw):
            break
        new, used = _rename_numpy_globals(raw[pos:])
        out.append(new)
        pos += used
    out.append(raw[pos:])
    return io.BytesIO(b"".join(out))


def read_checkpoint(path, device=None):
    """A checkpoint file is only ever read with ``weights_only=True`` (tensors, containers, plain numbers --
    nothing in the file is executed); files whose ``optim_dict`` holds numpy scalars (the reference's ``lr`` comes from
    ``np.load(lr.npy) * size``) get exactly those reconstructors allowed -- under the module name this numpy
    uses AND, for files written under the other numpy generation (``numpy.core`` vs ``numpy._core``), after renaming the
    module text in the pickle stream of an in-memory copy; a file that still does not load is refused."""
    if not os.path.exists(path):
        raise FileNotFoundError("File doesn't exist {}".format(path))
    where = device if device is not None else "cpu"
    try:
        return torch.load(path, map_location=where, weights_only=True)
    except Exception:
        with torch.serialization.safe_globals(_numpy_scalar_globals()):
            try:
                return torch.load(path, map_location=where, weights_only=True)
            except Exception:
                return torch.load(_renamed_copy(path), map_location=where, weights_only=True)


def load_checkpoint(checkpoint, model, optimizer=None, device=None, ismpi=False):
    """nnutils.py:129-151: restore ``model`` (and ``optimizer`` when given) from a file."""
    ck = read_checkpoint(checkpoint, "cpu")
    model.load_state_dict(ck["mpi_state_dict"] if ismpi else ck["state_dict"])
    if optimizer is not None and hasattr(optimizer, "load_state_dict") and "optim_dict" in ck:
        optimizer.load_state_dict(ck["optim_dict"])
    return ck
