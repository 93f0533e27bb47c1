"""BASELINE configs[2]-shaped end-to-end problem with a closed-form posterior, shared by the probe tool and the test:
nin = 26 parameters, nout = 457 data points, ``theory(theta) = A theta + c``, dense SPD covariance, flat priors wide
enough that truncation is > 8 sigma away, parameters 0 and 1 positive (the emulator sees their log10, as
``cosmolike_run.py:184,320`` sets ``dolog10index=[0, 1]``).  With noise-free data the posterior is Gaussian,
mean = theta_true, covariance = (A^T Sigma^-1 A)^-1."""
import numpy as np

NIN, NOUT = 26, 457
SEED = 97


def problem(seed=7, scale=0.08):
    rs = np.random.RandomState(seed)
    A = scale * rs.standard_normal((NOUT, NIN))
    for j in range(1, NIN):                                  # degenerate neighbours: posterior correlations up to ~0.7
        A[:, j] += 0.8 * A[:, j - 1] * (1 if j % 3 else -1)
    c = rs.uniform(size=NOUT)
    q, _ = np.linalg.qr(rs.standard_normal((NOUT, NOUT)))
    cov = (q * (np.logspace(0, -2, NOUT) * 0.1)[None, :]) @ q.T
    cov = 0.5 * (cov + cov.T)
    theta = rs.uniform(-0.3, 0.3, NIN)
    theta[0], theta[1] = 1.2, 0.9
    lo, hi = np.full(NIN, -1.0), np.full(NIN, 1.0)
    lo[:2], hi[:2] = 0.5, 2.0
    priors = [{"param": "p%d" % i, "dist": "flat", "arg1": float(lo[i]), "arg2": float(hi[i])} for i in range(NIN)]
    data = A @ theta + c
    icov = np.linalg.inv(cov)
    P = np.linalg.inv(A.T @ icov @ A)
    init = theta + 0.02 * rs.standard_normal(NIN)
    return dict(A=A, c=c, cov=cov, icov=icov, theta=theta, data=data, priors=priors, post_cov=0.5 * (P + P.T), init=init, lo=lo, hi=hi)


class Theory(object):
    """``theory(x, outdir)`` with ``x = (index, params)`` (util.py:763-770); picklable."""

    def __init__(self, A, c):
        self.A, self.c = A, c

    def __call__(self, x, outdir):
        return self.A @ np.asarray(x[1], np.float64) + self.c


def summary(chain, prob):
    """(max |mean - truth| / sigma, max |std / sigma - 1|, max |corr - corr_exact|)."""
    P = prob["post_cov"]
    sig = np.sqrt(np.diag(P))
    bias = np.abs(chain.mean(0) - prob["theta"]) / sig
    sd = chain.std(0) / sig
    corr = np.corrcoef(chain.T)
    corr_exact = P / np.outer(sig, sig)
    return float(bias.max()), float(np.abs(sd - 1).max()), float(np.abs(corr - corr_exact).max())
