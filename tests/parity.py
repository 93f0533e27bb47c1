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


# The numeric tolerances written in tests/test_gpu_*.py are 20x the error measured in round 4 on ONE box with ONE compiler
# (profiles/r04_parity_measured.json).  What the GPU suite ASSERTS is BOX_MARGIN x that = 50x the measured error -- the floor
# the smoke gate has: another box, driver or hipcc moves fp32 summation orders, not the algorithm.  The
# report records the tolerance as written (so the table stays comparable between rounds).
BOX_MARGIN = 2.5


def _gpu_site():
    f = sys._getframe(2)
    while f is not None and (f.f_code.co_filename.endswith("parity.py") or "numpy" in f.f_code.co_filename):
        f = f.f_back
    return f is not None and os.path.basename(f.f_code.co_filename).startswith("test_gpu_")


def assert_allclose(actual, desired, rtol=1e-7, atol=0, *args, **kw):
    if REPORT:
        _record("allclose", actual, desired, rtol, atol)
    if _gpu_site():
        rtol, atol = BOX_MARGIN * rtol, BOX_MARGIN * np.asarray(atol)
    return _orig(actual, desired, rtol, atol, *args, **kw)


def rowmax_close(got, ref, tol, floor=0.0, err_msg=""):
    """Gradient rows: every element within ``tol`` x the largest |ref| of its row (+ ``floor``).  The metric for
    d lnP / d z and parameter gradients: an element that is small next to its row's largest carries the row's absolute
    rounding error, not its own relative one."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    scale = np.abs(ref).max(axis=-1, keepdims=True)
    if REPORT:
        _record("rowmax", got, ref, tol, floor, scale=scale)
    if _gpu_site():
        tol, floor = BOX_MARGIN * tol, BOX_MARGIN * floor
    bad = np.abs(got - ref) > tol * scale + floor
    assert not np.any(bad), "%s%d of %d elements beyond %.1e x row max (worst %.2e)" % (
        err_msg + ": " if err_msg else "", int(bad.sum()), bad.size, tol, float(np.max(np.abs(got - ref) / (scale + 1e-300))))


def install():
    if np.testing.assert_allclose is not assert_allclose:
        np.testing.assert_allclose = assert_allclose


def near_relu_kink(z_row, emu, priors, radius=8 * 2.0 ** -24):
    """THE ReLU-kink exception of the gradient comparisons (one formula).  A row is exempt iff some hidden unit of that
    row has a pre-activation within fp32 rounding of zero:

        min over hidden units of  |a| / (sum_k |w_k x_k| + |b|)  <=  8 eps32        (a = w . x + b, float64 oracle)

    -- the sum's rounding error is a few eps32 of the magnitude sum, so which side of zero the unit falls on depends on
    the order of summation: the unit is on in one fp32 implementation and off in another (or in float64), and the row's
    gradient differs by a finite amount while lnP agrees to 1e-7.  Typical rows have a minimum of 2e-5 ... 5e-5 over the
    ~3000 hidden units of a ChtoModelv2; about one row in a thousand falls below 5e-7."""
    from oracle import likelihood, emulator
    x = likelihood.x_transform(likelihood.prior_map(np.asarray(z_row, np.float64)[None, :], priors), emu.X_mean, emu.X_std, emu.dolog10index)
    w = {k: np.asarray(v, np.float64) for k, v in emu.params.items()}
    h, worst = np.asarray(x, np.float64), np.inf

    def unit(pre, mag):
        nonlocal worst
        worst = min(worst, float((np.abs(pre) / np.maximum(mag, 1e-300)).min()))

    for op in emulator.topology(emu.kind, emu.in_size, emu.out_size, **emu.topo_kw):
        if op[0] == "linear":
            _, key, K, N, relu = op
            W, b = w[key + ".weight"], w[key + ".bias"]
            pre = h @ W.T + b
            if relu:
                unit(pre, np.abs(h) @ np.abs(W).T + np.abs(b))
                h = np.maximum(pre, 0.0)
            else:
                h = pre
        elif op[0] == "resblock":
            _, key, K, C, N = op
            W1, b1, W2, b2 = w[key + ".layer1.weight"], w[key + ".layer1.bias"], w[key + ".layer2.weight"], w[key + ".layer2.bias"]
            p1 = h @ W1.T + b1
            unit(p1, np.abs(h) @ np.abs(W1).T + np.abs(b1))
            t = np.maximum(p1, 0.0)
            Ws = w.get(key + ".skip_layer.weight")
            skip, mskip = (h @ Ws.T, np.abs(h) @ np.abs(Ws).T) if (Ws is not None and K != N) else (h, np.abs(h))
            p2 = (t @ W2.T + b2) * 0.1 + skip
            unit(p2, (np.abs(t) @ np.abs(W2).T + np.abs(b2)) * 0.1 + mskip)
            h = np.maximum(p2, 0.0)
        else:                                   # input skip: linear in the input, no gate
            break
    return worst <= radius
