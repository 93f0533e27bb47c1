"""The 33-D Gaussian of the reference's README (README.rst:60-88; BASELINE configs[0]), seeded: shared by the golden
generator and the tests.  The README draws ``init``, ``means``, ``cov`` in this order and does not seed."""
import numpy as np

SEED = 4321          # torch.manual_seed before the network is constructed
LR = 1e-3            # lr.npy (the range test needs torch_lr_finder, absent here: fixed instead)


def problem(seed=0):
    rs = np.random.RandomState(seed)
    ndim = 33
    init = rs.uniform(size=ndim)
    means = rs.uniform(size=ndim)
    cov = np.diag(0.1 * rs.uniform(size=ndim))
    priors = [{"param": "test_%d" % i, "dist": "flat", "arg1": -5.0, "arg2": 5.0} for i in range(ndim)]
    return dict(ndim=ndim, init=init, means=means, cov=cov, priors=priors)


def theory(x, outdirs):
    return np.array(x[1], copy=True)


def design(n, ndim, lo=-5.0, hi=5.0, seed=123456):
    """The point design ``train33_run.npz`` was generated on: a centred Latin hypercube with one ``permutation(n)`` per
    column.  Kept here, frozen, as an INPUT of that golden run -- the product's ``gensample_flat`` has since become the
    exact restatement of pyDOE2's design (pinned by the reference's own fixture), which orders the cells differently."""
    rs = np.random.RandomState(seed)
    u = np.stack([(rs.permutation(int(n)) + 0.5) / int(n) for _ in range(ndim)], axis=1)
    return lo + u * (hi - lo)
