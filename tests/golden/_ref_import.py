"""Import helper for the golden-vector generator (runs only in the build container).

The reference package at /root/reference imports several third-party packages that are
not installed here (emcee, zeus, h5py, pyDOE2, sample_generator, numdifftools, mpi4py,
schwimmbad, torch_lr_finder).  None of them is on the emulator hot path, so inert stub
modules are registered for exactly those names before `import linna`.  This file holds no
reference source; it only arranges sys.modules/sys.path.
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    class _Empty(object):
        def __init__(self, *a, **k):
            pass

    emcee = _stub("emcee")
    emcee.moves = _stub("emcee.moves", Move=_Empty)
    emcee.backends = _stub("emcee.backends", HDFBackend=_Empty)
    emcee.state = _stub("emcee.state", State=_Empty)
    emcee.EnsembleSampler = _Empty
    zeus = _stub("zeus")
    zeus.callbacks = _stub("zeus.callbacks", SaveProgressCallback=_Empty)
    zeus.autocorr = _stub("zeus.autocorr", AutoCorrTime=lambda *a, **k: None)
    zeus.EnsembleSampler = _Empty
    _stub("h5py")
    _stub("pyDOE2")
    _stub("sample_generator")
    _stub("numdifftools")
    _stub("torch_lr_finder", LRFinder=_Empty)
    # mpi4py / schwimmbad are wrapped in try/except by the reference itself.


def import_reference():
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import linna.nn as rnn
    import linna.util as rutil
    import linna.predictor_gpu as rpred
    import linna.HMCSampler as rhmc
    return rnn, rutil, rpred, rhmc
