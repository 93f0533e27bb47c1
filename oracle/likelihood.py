"""Oracle: prior map, input/output transforms, Gaussian log-likelihood (numpy).

TEST INFRASTRUCTURE ONLY.  Restates linna/util.py:291-381 (Transform), :466-596 (X/Y
transforms), :953-1021 (gaussianlogliklihood, Log_prob), :1160-1165 (lnprior) and
linna/predictor_gpu.py:461-504 (Predictor.predict).
"""
import numpy as np
from scipy.special import erf, erfinv

from . import emulator

SQRT2 = np.sqrt(2.0)


# ----------------------------------------------------------------------- prior map
def prior_arrays(priors):
    """Flatten the ``priors`` list (main.py:122-129) into (is_flat, arg1, arg2)."""
    is_flat = np.array([p["dist"] != "gauss" for p in priors])   # util.py:340-343
    a1 = np.array([p["arg1"] for p in priors], np.float64)
    a2 = np.array([p["arg2"] for p in priors], np.float64)
    return is_flat, a1, a2


def prior_map(z, priors):
    """util.py:339-343: gauss -> z*arg2+arg1; else Phi(z)*(arg2-arg1)+arg1."""
    z = np.atleast_2d(z)
    dt = z.dtype
    is_flat, a1, a2 = prior_arrays(priors)
    # torch: x / np.sqrt(2) with x float32 stays float32 (util.py:300)
    phi = dt.type(0.5) * (dt.type(1) + erf(z / dt.type(SQRT2)).astype(dt))
    flat = phi * (a2 - a1).astype(dt) + a1.astype(dt)
    gauss = z * a2.astype(dt) + a1.astype(dt)
    return np.where(is_flat[None, :], flat, gauss).astype(dt)


def prior_map_grad(z, priors):
    """d theta / d z, column-wise (derivative of ``prior_map``)."""
    z = np.atleast_2d(z)
    dt = z.dtype
    is_flat, a1, a2 = prior_arrays(priors)
    pdf = np.exp(-0.5 * z * z) / np.sqrt(2 * np.pi)
    return np.where(is_flat[None, :], pdf * (a2 - a1), a2[None, :] * np.ones_like(z)).astype(dt)


def inv_prior_map(theta, priors):
    """util.py:373-377 (invTransform): used once on ``init`` (main.py:132)."""
    theta = np.atleast_2d(theta)
    dt = theta.dtype
    is_flat, a1, a2 = prior_arrays(priors)
    u = (theta - a1.astype(dt)) / (a2 - a1).astype(dt)
    flat = dt.type(SQRT2) * erfinv(dt.type(2) * u - dt.type(1)).astype(dt)
    gauss = (theta - a1.astype(dt)) / a2.astype(dt)
    return np.where(is_flat[None, :], flat, gauss).astype(dt)


# ----------------------------------------------------------------------- transforms
def x_transform(theta, X_mean, X_std, dolog10index=None):
    """util.py:483-497."""
    x = np.array(theta, copy=True)
    if dolog10index is not None:
        for i in dolog10index:
            x[:, i] = np.log10(theta[:, i])
    return (x - X_mean[None, :].astype(x.dtype)) / X_std[None, :].astype(x.dtype)


def y_transform(y, y_mean, y_std, ypositive=False):
    """util.py:532-542."""
    v = y * y_std[None, :].astype(y.dtype) + y_mean[None, :].astype(y.dtype)
    return np.exp(v) if ypositive else v


def y_invtransform(y, y_mean, y_std, ypositive=False):
    """util.py:567-571."""
    v = np.log(y) if ypositive else y
    return (v - y_mean[None, :].astype(y.dtype)) / y_std[None, :].astype(y.dtype)


# ----------------------------------------------------------------------- emulator
class Emulator(object):
    """Everything ``Predictor.predict`` + ``Y_invtransform_data`` needs, as arrays."""

    def __init__(self, kind, in_size, out_size, params, X_mean, X_std, y_mean, y_std, sigma,
                 dolog10index=None, ypositive=False, **topo_kw):
        self.kind, self.in_size, self.out_size = kind, in_size, out_size
        self.params = params
        self.X_mean = np.asarray(X_mean, np.float32)
        self.X_std = np.asarray(X_std, np.float32)
        self.y_mean = np.asarray(y_mean, np.float32)
        self.y_std = np.asarray(y_std, np.float32)
        self.sigma = np.asarray(sigma, np.float32)
        self.dolog10index = dolog10index
        self.ypositive = ypositive
        self.topo_kw = topo_kw

    def network(self, x, keep=False):
        return emulator.forward(self.params, x, self.kind, self.in_size, self.out_size,
                                keep=keep, **self.topo_kw)

    def predict(self, theta):
        """predictor_gpu.py:461-504 followed by util.py:457-458 (``* sigma``)."""
        theta = np.atleast_2d(theta)
        x = x_transform(theta, self.X_mean, self.X_std, self.dolog10index)
        h = self.network(x)
        y = y_transform(h, self.y_mean, self.y_std, self.ypositive)
        return y * self.sigma[None, :].astype(y.dtype)


# ----------------------------------------------------------------------- likelihood
def gaussian_loglike_rows(m, data, invcov):
    """Row-wise restatement of util.py:953-955: -0.5 * d_i invcov d_i^T for every row i.

    The reference's function returns element [0][0] of a BxB product and is only right
    for B = 1; for B = 1 both agree.
    """
    d = m - data[None, :].astype(m.dtype)
    return (d @ invcov.astype(m.dtype) * d).sum(-1) * m.dtype.type(-0.5)


def lnprior_rows(z):
    """util.py:1160-1165 applied per row."""
    return z.dtype.type(-0.5) * (z * z).sum(-1)


def log_prob(z, emu, priors, data, invcov, temperature=1.0, dtype=np.float32):
    """util.py:990-1021 for a batch of latent points ``z[B, nin]`` -> ``[B]``.

    like = loglike/T + lnprior(z); NaN -> -inf (util.py:1013-1016).
    """
    z = np.atleast_2d(np.asarray(z)).astype(dtype)
    theta = prior_map(z, priors)
    m = emu.predict(theta)
    ll = gaussian_loglike_rows(m, np.asarray(data, dtype), np.asarray(invcov, dtype))
    out = ll / dtype(temperature) + lnprior_rows(z)
    out = np.where(np.isnan(out), -np.inf, out)
    return out.astype(dtype)


def log_prob_per_walker(z, emu, priors, data, invcov, temperature=1.0):
    """Reference-faithful evaluation order: one walker at a time, batch 1 (util.py:990)."""
    z = np.atleast_2d(z)
    return np.array([log_prob(zi[None, :], emu, priors, data, invcov, temperature)[0] for zi in z],
                    np.float32)


def grad_log_prob(z, emu, priors, data, invcov, temperature=1.0, dtype=np.float32):
    """d lnP / d z for every row: what ``torch.autograd.grad(lnP, x)`` returns at
    HMCSampler.py:32 when lnP is ``Log_prob(nograd=False)``; the intended semantics of
    util.py:1023-1035 (``Dlnp``, broken as shipped: SURVEY §8 a17).

    Returns (lnP[B], grad[B, nin]).
    """
    z = np.atleast_2d(np.asarray(z)).astype(dtype)
    theta = prior_map(z, priors)
    x = x_transform(theta, emu.X_mean, emu.X_std, emu.dolog10index)
    h, caches = emu.network(x, keep=True)
    v = h * emu.y_std[None, :].astype(dtype) + emu.y_mean[None, :].astype(dtype)
    y = np.exp(v) if emu.ypositive else v
    m = y * emu.sigma[None, :].astype(dtype)
    data = np.asarray(data, dtype)
    invcov = np.asarray(invcov, dtype)
    d = m - data[None, :]
    ll = (d @ invcov * d).sum(-1) * dtype(-0.5)
    lnp = ll / dtype(temperature) + lnprior_rows(z)
    # backward
    dm = -(d @ invcov + d @ invcov.T) * dtype(0.5) / dtype(temperature)
    dy = dm * emu.sigma[None, :].astype(dtype)
    dv = dy * y if emu.ypositive else dy
    dh = dv * emu.y_std[None, :].astype(dtype)
    dx, _ = emulator.backward(emu.params, caches, dh, emu.kind, emu.in_size, emu.out_size,
                              need_param_grads=False, **emu.topo_kw)
    dtheta = dx / emu.X_std[None, :].astype(dtype)
    if emu.dolog10index is not None:
        for i in emu.dolog10index:
            dtheta[:, i] = dtheta[:, i] / (theta[:, i] * dtype(np.log(10.0)))
    dz = dtheta * prior_map_grad(z, priors) - z
    lnp = np.where(np.isnan(lnp), -np.inf, lnp)
    return lnp.astype(dtype), dz.astype(dtype)
