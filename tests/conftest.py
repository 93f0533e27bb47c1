import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import parity
    parity.install()        # LINNA_PARITY_REPORT=<file>: every assert_allclose records its measured worst error (tests/parity.py)


def pytest_sessionstart(session):
    """A fresh checkout has no liblinna_hip.so (built artefacts are git-ignored): build it once
    (hipcc cross-compiles gfx950 without a GPU)."""
    from linna_amd import _lib, _build
    if not os.path.exists(_lib.LIB_PATH):
        _build.build(verbose=False)


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
