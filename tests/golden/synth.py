"""Deterministic synthetic weights / inputs shared by make_golden.py and the tests.

Everything is drawn from ``np.random.RandomState`` (frozen legacy stream), so the weights
need not be stored in the fixtures: the generator loads them into the reference model and
the tests rebuild the same arrays.  No reference code here.
"""
import numpy as np


def _shapes(kind, nin, nout, width=512, depth=4):
    """state_dict key -> shape in torch's state_dict order (restated from nn.py:73-86)."""
    shapes = {}

    def lin(key, K, N, bias=True):
        shapes[key + ".weight"] = (N, K)
        if bias:
            shapes[key + ".bias"] = (N,)

    if kind == "MLP":
        k = nin
        for i in range(depth):
            lin("layer%d" % (i + 1), k, width)
            k = width
        lin("layer%d" % (depth + 1), k, nout)
        return shapes
    channel = 4 if kind == "ChtoModelsimple" else 16
    h = 1000 if nout > 30 else max(32, 32 * nout)
    lin("layer1", nin, h)
    for name, mult in (("layer2", 1), ("layer3", 2), ("layer4", 4)):
        c = channel * mult
        lin(name + ".layer1", h, c)
        lin(name + ".layer2", c, h // 2)
        lin(name + ".skip_layer", h, h // 2, bias=False)
        h //= 2
    h6 = h if kind == "ChtoModelsimple" else 4 * h
    lin("layer6", h, h6)
    lin("layer7", h6, nout)
    lin("layer8", nout, nout)
    if kind == "ChtoModelv2_linear":
        lin("linearlayer", nin, nout)
    return shapes


def weights(kind, nin, nout, seed, **kw):
    """He-like random weights (all tensors non-trivial, skip paths included)."""
    rs = np.random.RandomState(seed)
    p = {}
    for key, shp in _shapes(kind, nin, nout, **kw).items():
        if key.endswith("bias"):
            p[key] = (0.1 * rs.standard_normal(shp)).astype(np.float32)
        else:
            scale = np.sqrt(2.0 / shp[1])
            if "skip_layer" in key:
                scale *= 0.7
            p[key] = (scale * rs.standard_normal(shp)).astype(np.float32)
    return p


def gaussian_problem(nin, nout, seed, dense=False, cond=1e3):
    """data, cov, priors for an (nin -> nout) problem; README.rst:69-83 shaped when diag."""
    rs = np.random.RandomState(seed)
    data = rs.uniform(size=nout)
    if not dense:
        cov = np.diag(0.1 * rs.uniform(0.05, 1.0, size=nout))
    else:
        q, _ = np.linalg.qr(rs.standard_normal((nout, nout)))
        ev = np.logspace(0, -np.log10(cond), nout) * 0.1
        cov = (q * ev[None, :]) @ q.T
        cov = 0.5 * (cov + cov.T)
    priors = []
    for i in range(nin):
        if i % 3 == 2:
            priors.append({"param": "p%d" % i, "dist": "gauss", "arg1": 0.3, "arg2": 0.8})
        else:
            priors.append({"param": "p%d" % i, "dist": "flat", "arg1": -5.0, "arg2": 5.0})
    return data, cov, priors


def cond_problem(nin, nout, seed, cond):
    """Dense SPD covariance of a stated condition number WITH its inverse from the same eigen-decomposition (the generator
    and the tests then hold the same inverse to the last bits that survive fp32 rounding, whatever LAPACK's ``inv`` does
    at condition 1e6), and the covariance's symmetric square root (to draw residuals the covariance calls likely)."""
    rs = np.random.RandomState(seed)
    q, _ = np.linalg.qr(rs.standard_normal((nout, nout)))
    ev = np.logspace(0, -np.log10(cond), nout) * 0.1
    cov = (q * ev[None, :]) @ q.T
    inv = (q / ev[None, :]) @ q.T
    half = (q * np.sqrt(ev)[None, :]) @ q.T
    return 0.5 * (cov + cov.T), 0.5 * (inv + inv.T), half


def transform_constants(nin, nout, seed):
    rs = np.random.RandomState(seed + 7)
    X_mean = rs.uniform(-0.5, 0.5, nin).astype(np.float32)
    X_std = rs.uniform(0.5, 3.0, nin).astype(np.float32)
    y_mean = rs.uniform(-0.5, 0.5, nout).astype(np.float32)
    y_std = rs.uniform(0.5, 2.0, nout).astype(np.float32)
    return X_mean, X_std, y_mean, y_std


def latent_points(n, nin, seed):
    return np.random.RandomState(seed + 13).standard_normal((n, nin)).astype(np.float32)


def tensor_digest(a, nsamp=16, seed=0):
    """Small summary of a big tensor: [sum, sum|.|, sum(.^2)] + ``nsamp`` fixed entries."""
    a = np.asarray(a, np.float64).ravel()
    idx = np.random.RandomState(seed + a.size).randint(0, a.size, size=min(nsamp, a.size))
    return np.concatenate([[a.sum(), np.abs(a).sum(), (a * a).sum()], a[idx]])
