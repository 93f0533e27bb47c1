"""Measured parity errors next to the tolerances that assert them.

``LINNA_PARITY_REPORT=<file>`` makes every ``np.testing.assert_allclose`` of the test session (and the row-wise gradient
checks that go through ``rowmax_close``) append one JSON line: test id, call site, the tolerance asserted and the WORST
error measured in the same metric -- ``max |got - ref| / (atol + rtol |ref|)`` as a fraction of the tolerance, and the
effective relative error.  ``tools/parity_report.py`` folds the file into one table (worst per call site); the numeric
tolerances in tests/test_gpu_*.py are set from that table to at most 20x the measured error, and the
table of the round is committed as profiles/r04_parity_measured.json.
"""
import json
import os
import sys

import numpy as np

REPORT = os.environ.get("LINNA_PARITY_REPORT")
_orig = np.testing.assert_allclose


def _site():
    f = sys._getframe(2)
    while f is not None and (f.f_code.co_filename.endswith("parity.py") or "numpy" in f.f_code.co_filename):
        f = f.f_back
    return "%s:%d" % (os.path.basename(f.f_code.co_filename), f.f_lineno) if f is not None else "?"


def _record(kind, got, ref, rtol, atol, scale=None):
    try:
        g = np.asarray(got, np.float64)
        r = np.asarray(ref, np.float64)
        a = np.broadcast_to(np.asarray(atol, np.float64), np.broadcast(g, r).shape) if np.ndim(atol) else float(atol)
        fin = np.isfinite(g) & np.isfinite(r)
        if not np.any(fin):
            return
        d = np.abs(g - r)
        den = (a + rtol * np.abs(r)) if scale is None else (a + rtol * scale)
        with np.errstate(divide="ignore", invalid="ignore"):
            frac = np.where(fin & (den > 0), d / den, 0.0)
            rel = np.where(fin & (np.abs(r) > 0), d / np.abs(r), 0.0) if scale is None else np.where(fin, d / np.broadcast_to(scale, d.shape), 0.0)
        rec = {"test": os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], "site": _site(), "kind": kind,
               "rtol": float(rtol), "atol": float(np.max(atol)), "n": int(d.size), "worst_frac_of_tol": float(np.max(frac)),
               "max_rel_err": float(np.max(rel)), "max_abs_err": float(np.max(np.where(fin, d, 0.0)))}
        with open(REPORT, "a") as f:
            f.write(json.dumps(rec) + "\n")
    except Exception as e:                                   # noqa: BLE001  (the report must never fail a test)
        sys.stderr.write("parity report: %r\n" % (e,))


def assert_allclose(actual, desired, rtol=1e-7, atol=0, *args, **kw):
    if REPORT:
        _record("allclose", actual, desired, rtol, atol)
    return _orig(actual, desired, rtol, atol, *args, **kw)


def rowmax_close(got, ref, tol, floor=0.0, err_msg=""):
    """Gradient rows: every element within ``tol`` x the largest |ref| of its row (+ ``floor``).  The metric for
    d lnP / d z and parameter gradients: an element that is small next to its row's largest carries the row's absolute
    rounding error, not its own relative one."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    scale = np.abs(ref).max(axis=-1, keepdims=True)
    if REPORT:
        _record("rowmax", got, ref, tol, floor, scale=scale)
    bad = np.abs(got - ref) > tol * scale + floor
    assert not np.any(bad), "%s%d of %d elements beyond %.1e x row max (worst %.2e)" % (
        err_msg + ": " if err_msg else "", int(bad.sum()), bad.size, tol, float(np.max(np.abs(got - ref) / (scale + 1e-300))))


def install():
    if REPORT and np.testing.assert_allclose is not assert_allclose:
        np.testing.assert_allclose = assert_allclose
