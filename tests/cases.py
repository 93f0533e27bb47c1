"""Problem construction shared by the tests: rebuilds, WITHOUT the reference, exactly the
synthetic problems tests/golden/make_golden.py fed to the reference."""
import os

import numpy as np

import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SERVING = [
    ("v2_33_33", "ChtoModelv2", 33, 33, 101, False, 64, None, False, {}),
    ("mlp_33_33", "MLP", 33, 33, 102, False, 64, None, False, {}),
    ("mlp_33_33_dense", "MLP", 33, 33, 103, True, 64, None, False, {}),
    ("v2_26_457", "ChtoModelv2", 26, 457, 104, True, 24, None, False, {}),
    ("v2_40_1000", "ChtoModelv2", 40, 1000, 105, True, 12, None, False, {}),
    ("simple_6_4", "ChtoModelsimple", 6, 4, 106, True, 64, None, False, {}),
    ("v2lin_5_3_log10", "ChtoModelv2_linear", 5, 3, 107, True, 64, [0, 1], False, {}),
    ("v2_4_2_ypos", "ChtoModelv2", 4, 2, 108, False, 64, None, True, {}),
    ("mlp_7_5_small", "MLP", 7, 5, 109, True, 64, None, False, {"width": 48, "depth": 3}),
]
SERVING_BY_NAME = {c[0]: c for c in SERVING}

TRAIN = [
    ("train_v2_5_3", "ChtoModelv2", 5, 3, 201, 40, {}, True),
    ("train_mlp_7_5", "MLP", 7, 5, 202, 40, {"width": 48, "depth": 3}, True),
    ("train_v2_33_33", "ChtoModelv2", 33, 33, 203, 100, {}, False),
    ("train_v2_12_40", "ChtoModelv2", 12, 40, 204, 50, {}, False),
    ("train_v2_26_457", "ChtoModelv2", 26, 457, 205, 64, {}, False),      # BASELINE config 3 shape
    ("train_v2lin_5_3", "ChtoModelv2_linear", 5, 3, 206, 40, {}, True),
    ("train_simple_6_4", "ChtoModelsimple", 6, 4, 207, 40, {}, True),
]


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def serving_problem(name):
    """dict(kind, nin, nout, kw, weights, priors, data, cov, invcov, sigma, X_mean, X_std,
    y_mean, y_std, dolog10, ypositive) for a SERVING case."""
    _, kind, nin, nout, seed, dense, n, dolog10, ypos, kw = SERVING_BY_NAME[name]
    data, cov, priors = synth.gaussian_problem(nin, nout, seed, dense=dense)
    if dolog10 is not None:
        for i in dolog10:
            priors[i] = {"param": "p%d" % i, "dist": "flat", "arg1": 0.1, "arg2": 2.0}
    if ypos:
        data = np.abs(data) + 0.5
    X_mean, X_std, y_mean, y_std = synth.transform_constants(nin, nout, seed)
    if ypos:
        y_std = (0.1 * y_std).astype(np.float32)
    return dict(kind=kind, nin=nin, nout=nout, kw=kw, weights=synth.weights(kind, nin, nout, seed, **kw),
                priors=priors, data=data, cov=cov, invcov=np.linalg.inv(cov), sigma=np.sqrt(np.diag(cov)),
                X_mean=X_mean, X_std=X_std, y_mean=y_mean, y_std=y_std, dolog10=dolog10, ypositive=ypos)


def oracle_emulator(prob):
    from oracle.likelihood import Emulator
    return Emulator(prob["kind"], prob["nin"], prob["nout"], prob["weights"], prob["X_mean"], prob["X_std"],
                    prob["y_mean"], prob["y_std"], prob["sigma"], dolog10index=prob["dolog10"],
                    ypositive=prob["ypositive"], **prob["kw"])


def training_problem(name):
    _, kind, nin, nout, seed, B, kw, full = [c for c in TRAIN if c[0] == name][0]
    rs = np.random.RandomState(seed + 31)
    data, cov, _ = synth.gaussian_problem(nin, nout, seed, dense=True, cond=1e2)
    sigma = np.sqrt(np.diag(cov))
    X_mean, X_std, y_mean, y_std = synth.transform_constants(nin, nout, seed)
    X = (X_mean[None, :] + X_std[None, :] * rs.standard_normal((3, B, nin))).astype(np.float32)
    Y = (data[None, None, :] + 3 * sigma[None, None, :] * rs.standard_normal((3, B, nout))).astype(np.float32)
    Y[0, 1, 0] = 1e10
    Y[1, 2, nout - 1] = 1e-30
    return dict(kind=kind, nin=nin, nout=nout, kw=kw, full=full, weights=synth.weights(kind, nin, nout, seed, **kw),
                data=data, cov=cov, sigma=sigma, X_mean=X_mean, X_std=X_std, y_mean=y_mean, y_std=y_std, X=X, Y=Y)
