"""Likelihood, transforms and model retrieval of LINNA on MI355X.

Mirrors the hot-path part of the reference's ``linna/util.py`` (same class names and
constructor signatures) so that code written against ``linna.util`` keeps working:
``Transform``/``invTransform`` (util.py:313-381), ``X_transform_class`` (:466-510),
``Y_transform_class``/``Y_invtransform_class`` (:512-596), ``Y_transform_data``/
``Y_invtransform_data`` (:402-464), ``gaussianlogliklihood`` (:953-955), ``Log_prob``
(:957-1021), ``lnprior`` (:1160-1165), ``retrieve_model`` (:611-639).

The small transform classes are containers of constants (their ``__call__`` is kept for
API parity and operates on whatever tensor it is given); the hot path -- ``Log_prob`` on
a batch of walkers -- runs as ONE fused pipeline of HIP kernels (``linna_logprob_eval``).
"""
import ctypes as C
import io
import os
import pickle
from collections import OrderedDict
from copy import deepcopy

import numpy as np
import torch

from . import _lib
from . import nn as lnn
from . import predictor_gpu
from .nn import *  # noqa: F401,F403  (the reference re-exports its network classes here)

SQRT2 = float(np.sqrt(2.0))


def cpu_quota():
    """CPUs this process may use: the scheduler affinity, capped by the cgroup bandwidth quota (cpu.max / cfs_quota_us)
    when there is one -- a container on a 256-core MI355X host typically has 16."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()))):
        try:
            quota, period = parse(open(path).read())
            if quota not in ("max", "-1") and float(quota) > 0:
                n = min(n, max(1, int(float(quota) / float(period))))
            break
        except Exception:
            continue
    return n


def limit_threads_to_quota():
    """torch sizes its intra-op thread pool by the host's core count; with more threads than the container's CPU quota
    every parallel CPU op over 32768 elements exhausts the quota within the scheduling period and the whole process --
    the thread feeding the GPU included -- is frozen for the rest of it (measured: 65 ms of every 100).  Called by
    ``ml_sampler_core``; lowers ``torch.get_num_threads()`` to the quota, never raises it."""
    q = cpu_quota()
    if torch.get_num_threads() > q:
        torch.set_num_threads(q)
    return torch.get_num_threads()


# ------------------------------------------------------------------ prior map (util.py:291-381)
def gauss2unif(x):
    return 0.5 * (1 + torch.erf(x / SQRT2))


def invgauss2unif(x):
    return SQRT2 * torch.erfinv(2 * x - 1)


def _as_2d_tensor(x, inputnumpy):
    if inputnumpy:
        x = torch.from_numpy(np.asarray(x).astype(np.float32))
    if x.dim() < 2:
        x = x.reshape(-1, len(x))
    return x


class Transform(object):
    """latent z (unit Gaussian per parameter) -> physical parameter theta (util.py:313-347)."""

    def __init__(self, priors):
        self.priors = priors

    def arrays(self):
        """(is_flat int32[n], a1 float32[n], a2 float32[n]) as consumed by linna_prior_map_fwd;
        for flat priors a2 is the width arg2-arg1 (the scalar the reference multiplies by)."""
        is_flat = np.array([0 if p["dist"] == "gauss" else 1 for p in self.priors], np.int32)
        a1 = np.array([p["arg1"] for p in self.priors], np.float32)
        a2 = np.array([p["arg2"] if p["dist"] == "gauss" else (p["arg2"] - p["arg1"]) for p in self.priors], np.float32)
        return is_flat, a1, a2

    def __call__(self, x, returnnumpy=True, inputnumpy=True):
        x = _as_2d_tensor(x, inputnumpy)
        cols = []
        for i, p in enumerate(self.priors):
            if p["dist"] == "gauss":
                cols.append(x[:, i] * p["arg2"] + p["arg1"])
            else:
                cols.append(gauss2unif(x[:, i]) * (p["arg2"] - p["arg1"]) + p["arg1"])
        out = torch.stack(cols).T.squeeze()
        return out.detach().cpu().numpy() if returnnumpy else out


class invTransform(object):
    """theta -> z (util.py:349-381); used once on ``init`` (main.py:132)."""

    def __init__(self, priors):
        self.priors = priors

    def __call__(self, x, returnnumpy=True, inputnumpy=True):
        x = _as_2d_tensor(x, inputnumpy)
        cols = []
        for i, p in enumerate(self.priors):
            if p["dist"] == "gauss":
                cols.append((x[:, i] - p["arg1"]) / p["arg2"])
            else:
                cols.append(invgauss2unif((x[:, i] - p["arg1"]) / (p["arg2"] - p["arg1"])))
        out = torch.stack(cols).T.squeeze()
        return out.detach().cpu().numpy() if returnnumpy else out


# ------------------------------------------------------------------ data-vector transforms
class _Picklable(object):
    def pickle(self, path):
        """The reference's ``pickle()`` methods (util.py:424, 464, 499, 544, 592): a CPU copy of the object.  The stream names
        the class as ``linna.util.<Class>`` -- same attributes as the reference's class of that name -- so that a run
        directory written here continues under the reference as well (its unpickler finds its own class); this package's
        ``CPU_Unpickler`` maps either module name to the classes here."""
        with open(path, "wb") as f:
            new = deepcopy(self)
            new.dev = "cpu"
            for k, v in list(new.__dict__.items()):
                if torch.is_tensor(v):
                    new.__dict__[k] = v.detach().cpu()
            data = pickle.dumps(new, 3)                 # protocol 3: the class travels as a GLOBAL text record, bytes natively, no frames
            tag = ("c%s\n%s\n" % (type(self).__module__, type(self).__name__)).encode()
            assert data.count(tag) == 1
            f.write(data.replace(tag, ("clinna.util\n%s\n" % type(self).__name__).encode()))


class Y_transform_data(_Picklable):
    """y -> y / sigma (util.py:402-447)."""

    def __init__(self, sigma, device="cpu"):
        self.device = device
        self.sigma = torch.from_numpy(np.asarray(sigma).astype(np.float32)).to(device)

    def __call__(self, y):
        return y / self.sigma[None, :].to(y.device)

    def transform_cov(self, cov):
        d = torch.diag(1 / self.sigma.detach().cpu().type(torch.float64))
        return d.inner(torch.as_tensor(cov, dtype=torch.float64).cpu()).inner(d)


class Y_invtransform_data(_Picklable):
    """y -> y * sigma (util.py:449-464)."""

    def __init__(self, sigma, device="cpu"):
        self.device = device
        self.sigma = torch.from_numpy(np.asarray(sigma).astype(np.float32)).to(device)

    def __call__(self, y):
        return y * self.sigma[None, :].to(y.device)


class X_transform_class(_Picklable):
    """x -> (x - mean)/std, log10 first on ``dolog10index`` columns (util.py:466-497)."""

    def __init__(self, X_mean, X_std, device="cpu", dolog10index=None):
        # the reference documents dolog10index as an "int array": kept as a plain list of ints, so that X_transform.pkl
        # holds nothing but builtins, tensors and this class (what CPU_Unpickler reads back; the reference iterates it)
        if dolog10index is not None:
            dolog10index = [int(i) for i in np.asarray(dolog10index).reshape(-1)]
        self.X_mean, self.X_std, self.dev, self.dolog10index = X_mean, X_std, device, dolog10index

    def __call__(self, X):
        X1 = X.clone()
        if self.dolog10index is not None:
            for ind in self.dolog10index:
                if X1.dim() > 1:
                    X1[:, ind] = torch.log10(X[:, ind])
                else:
                    X1[ind] = torch.log10(X1[ind])
        return (X1 - self.X_mean[None, :].to(X.device)) / self.X_std[None, :].to(X.device)


class Y_transform_class(_Picklable):
    """network space -> sigma units: y*std+mean, exp(.) if ``ypositive`` (util.py:512-542)."""

    def __init__(self, y_mean, y_std, dev="cpu", ypositive=False):
        self.y_mean, self.y_std, self.dev, self.ypositive = y_mean, y_std, dev, ypositive

    def __call__(self, y):
        v = y * self.y_std[None, :].to(y.device) + self.y_mean[None, :].to(y.device)
        return torch.exp(v) if self.ypositive else v


class Y_invtransform_class(_Picklable):
    """inverse of ``Y_transform_class`` (util.py:556-590)."""

    def __init__(self, y_mean, y_std, data_tensor, dev="cpu", ypositive=False):
        self.y_mean, self.y_std, self.dev, self.ypositive, self.data_tensor = y_mean, y_std, dev, ypositive, data_tensor

    def __call__(self, y):
        v = torch.log(y) if self.ypositive else y
        return (v - self.y_mean[None, :].to(y.device)) / self.y_std[None, :].to(y.device)

    def transform_cov(self, cov):
        cov = torch.as_tensor(cov, dtype=torch.float64).cpu()
        s = torch.diag(1 / self.y_std.detach().cpu().type(torch.float64))
        if self.ypositive:
            e = torch.diag(1 / self.data_tensor.detach().cpu().type(torch.float64))
            cov0 = e.inner(cov).inner(e)
            cov0[cov0 <= -1] = 1e-10 - 1
            cov = torch.log(1 + cov0)
        return s.inner(cov).inner(s)


# ------------------------------------------------------------------ reading the reference's artefacts
_REFERENCE_CLASSES = ("Transform", "invTransform", "Y_transform_data", "Y_invtransform_data",
                      "X_transform_class", "Y_transform_class", "Y_invtransform_class")


class CPU_Unpickler(pickle.Unpickler):
    """util.py:51-55 as a CLOSED allow-list: the transform pickles of a run directory name the seven
    transform classes (``linna.util.*`` when the reference wrote them: loaded as the classes of this
    module, same attribute names), ``collections.OrderedDict``, ``torch._utils._rebuild_tensor_v2`` and
    ``torch.storage._load_from_bytes``; the storage bytes go through ``torch.load(weights_only=True)``.
    Any other global is refused -- nothing a file names is ever imported or called."""

    def find_class(self, module, name):
        if module == "torch.storage" and name == "_load_from_bytes":
            return _storage_from_bytes
        if module == "torch._utils" and name == "_rebuild_tensor_v2":
            return torch._utils._rebuild_tensor_v2
        if module == "collections" and name == "OrderedDict":
            return OrderedDict
        if module in ("linna.util", "linna_amd.util") and name in _REFERENCE_CLASSES:
            return globals()[name]
        # data-only numpy reconstructors (a transform pickle the REFERENCE wrote keeps dolog10index / sigma as the numpy
        # arrays it was given), under either numpy generation's module name -- the same four ArgsUnpickler allows
        core = getattr(np, "_core", None) or np.core
        if module in ("numpy.core.multiarray", "numpy._core.multiarray") and name in ("_reconstruct", "scalar"):
            return getattr(core.multiarray, name)
        if module == "numpy" and name in ("ndarray", "dtype"):
            return getattr(np, name)
        raise pickle.UnpicklingError("refusing to load global %s.%s from a transform pickle" % (module, name))


def _storage_from_bytes(b):
    return torch.load(io.BytesIO(b), map_location="cpu", weights_only=True)


class ArgsUnpickler(pickle.Unpickler):
    """``model_args.pkl`` (main.py:189-198: the positional arguments of ``train_NN``) read with a closed allow-list:
    numpy array / scalar reconstructors, the point-design helper and the network classes of ``linna.nn`` /
    ``linna.util`` (loaded as this package's), and ``train_NN`` itself (``model_pickle.pkl``)."""

    def find_class(self, module, name):
        core = getattr(np, "_core", None) or np.core
        if module in ("numpy.core.multiarray", "numpy._core.multiarray") and name in ("_reconstruct", "scalar"):
            return getattr(core.multiarray, name)
        if module == "numpy" and name in ("ndarray", "dtype"):
            return getattr(np, name)
        if module in ("linna.util", "linna_amd.util") and name in ("NN_samplerv1", "train_NN"):
            return globals()[name]
        if module in ("linna.nn", "linna_amd.nn", "linna.util", "linna_amd.util") and name in lnn.__all__:
            return getattr(lnn, name)
        raise pickle.UnpicklingError("refusing to load global %s.%s from model_args.pkl" % (module, name))


def retrieve_model(outdir, inshape, outshape, nnmodel_in=lnn.ChtoModelv2, device=None):
    """Rebuild the trained emulator from ``outdir`` (util.py:611-639): ``best.pth.tar`` plus the
    three transform pickles, in the reference's on-disk formats."""
    with open(os.path.join(outdir, "y_invtransform_data.pkl"), "rb") as f:
        y_invtransform_data = CPU_Unpickler(f).load()
    with open(os.path.join(outdir, "X_transform.pkl"), "rb") as f:
        X_transform = CPU_Unpickler(f).load()
    X_transform.dev = "cpu"
    with open(os.path.join(outdir, "y_transform.pkl"), "rb") as f:
        y_transform = CPU_Unpickler(f).load()
    y_transform.dev = "cpu"
    nnmodel = nnmodel_in(inshape, outshape, None)
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    model = predictor_gpu.Predictor(inshape, outshape, X_transform=X_transform, y_transform=y_transform,
                                    device=device, outdir=outdir, model=nnmodel)
    model.load_checkpoint()
    return model, y_invtransform_data


# ------------------------------------------------------------------ likelihood
def gaussianlogliklihood(m, data, invcov):
    """util.py:953-955 (per walker, m is [1, nout]).  Kept for callers that pass it around;
    ``Log_prob`` recognises it and runs the fused HIP pipeline instead."""
    d = m - data
    return (d @ invcov @ d.T * (-0.5))[0][0]


def lnprior(x):
    """util.py:1160-1165."""
    return -0.5 * torch.sum(x.square())


def _np32(t):
    if torch.is_tensor(t):
        return t.detach().cpu().numpy().astype(np.float32)
    return np.asarray(t, np.float32)


class Log_prob(object):
    """Posterior log-probability of latent walker positions (util.py:957-1021).

    ``__call__`` accepts one walker ``x[ndim]`` (reference semantics: returns a scalar) or a
    batch ``x[B, ndim]`` (returns ``[B]``, every row evaluated as the reference evaluates one
    walker -- the reference itself is only correct for B = 1).
    """

    def __init__(self, data_new, invcov_new, model, y_invtransform_data, transform, temperature, loglikelihoodfunc=None,
                 nograd=False, externalloglike=None):
        self.data_new = data_new
        self.invcov_new = invcov_new
        self.model = model
        self.y_invtransform_data = y_invtransform_data
        self.transform = transform
        self.T = float(temperature)
        self.no_grad = nograd
        self.loglikelihoodfunc = loglikelihoodfunc if loglikelihoodfunc is not None else gaussianlogliklihood
        self.noduplicate = True            # consumed by the reference's MPI pool (util.py:987)
        self.externalloglike = externalloglike
        self._plan = None
        self._ws = {}

    # -------------------------------------------------------------- device plan
    def _build(self):
        pred = self.model
        net = pred.model
        dev = net.device
        if dev.type != "cuda":
            raise _lib.LinnaHipError("Log_prob needs the emulator on the GPU (no CPU fallback)")
        nin, nout = net.in_size, net.out_size
        # (a private writable copy: torch warns on read-only arrays, e.g. views of an .npz member)
        f32 = lambda a: torch.as_tensor(np.array(a, dtype=np.float32, order="C", copy=True), device=dev)
        i32 = lambda a: torch.as_tensor(np.array(a, dtype=np.int32, order="C", copy=True), device=dev)
        is_flat, a1, a2 = self.transform.arrays()
        if len(is_flat) != nin:
            raise ValueError("%d priors for a %d-input emulator" % (len(is_flat), nin))
        Xt, Yt = pred.X_transform, pred.y_transform
        lg = np.zeros(nin, np.int32)
        if getattr(Xt, "dolog10index", None) is not None:
            lg[list(Xt.dolog10index)] = 1
        sigma = _np32(self.y_invtransform_data.sigma).astype(np.float64)
        y_mean, y_std = _np32(Yt.y_mean).astype(np.float64), _np32(Yt.y_std).astype(np.float64)
        data = _np32(self.data_new).astype(np.float64)
        S = _np32(self.invcov_new)
        k = dict(is_flat=i32(is_flat), a1=f32(a1), a2=f32(a2), lg=i32(lg), xmean=f32(_np32(Xt.X_mean)),
                 xstd=f32(_np32(Xt.X_std)), gscale=f32(y_std * sigma))
        d = _lib.LogprobDesc()
        d.nin, d.nout = nin, nout
        d.is_flat, d.a1, d.a2 = _lib.iptr(k["is_flat"]), _lib.ptr(k["a1"]), _lib.ptr(k["a2"])
        d.log10_flag = _lib.iptr(k["lg"]) if lg.any() else None
        d.xmean, d.xstd = _lib.ptr(k["xmean"]), _lib.ptr(k["xstd"])
        if getattr(Yt, "ypositive", False):
            # m - data = exp(h*ystd + ymean)*sigma - data            (util.py:540, 458)
            k.update(cscale=f32(y_std), cshift=f32(y_mean), cpost=f32(sigma), cshift2=f32(-data))
            d.outmap.cexp = 1
            d.outmap.cpost, d.outmap.cshift2 = _lib.ptr(k["cpost"]), _lib.ptr(k["cshift2"])
        else:
            # m - data = h*(ystd*sigma) + (ymean*sigma - data)       (util.py:542, 458)
            k.update(cscale=f32(y_std * sigma), cshift=f32(y_mean * sigma - data))
        d.outmap.cscale, d.outmap.cshift = _lib.ptr(k["cscale"]), _lib.ptr(k["cshift"])
        if np.count_nonzero(S - np.diag(np.diagonal(S))) == 0:
            k["w"] = f32(np.diagonal(S))
            d.w = _lib.ptr(k["w"])
        else:
            # rows padded to a multiple of 4 floats so the MFMA GEMM streams them by 16-byte LDS-DMA
            ldS = _lib.ld4(nout)
            pad = lambda m: np.pad(m, ((0, 0), (0, ldS - nout)))
            k["S"] = f32(pad(S))
            k["Ssym"] = f32(pad(0.5 * (S.astype(np.float64) + S.astype(np.float64).T)))
            d.S, d.lds, d.Ssym = _lib.ptr(k["S"]), ldS, _lib.ptr(k["Ssym"])
            # lnP as |d L|^2 with S = L L^T (float64 Cholesky of the matrix the caller gave, i.e. of the reference's fp32
            # invcov): the same GEMM, without the cancellation d S d^T has when the covariance is ill-conditioned
            # (SURVEY 7 "hard parts"; measured in tests/test_gpu_cond.py).  Not positive definite: the direct form.
            if os.environ.get("LINNA_DENSE_FACTORED", "1") != "0":
                try:
                    Lf = np.linalg.cholesky(0.5 * (S.astype(np.float64) + S.astype(np.float64).T))
                    k["Sfac"] = f32(pad(Lf))
                    d.Sfac = _lib.ptr(k["Sfac"])
                except np.linalg.LinAlgError:
                    pass
        d.gscale = _lib.ptr(k["gscale"])
        d.temperature = self.T
        h = C.c_void_p()
        _lib.call("linna_logprob_create", _lib.ctx(dev.index), net.net_handle(), C.byref(d), C.byref(h))
        self._plan = dict(handle=h, keep=k, desc=d, dev=dev, nin=nin, nout=nout, net_sig=net._net_sig)
        self._ws = {}

    def _ensure(self):
        net = self.model.model
        if self._plan is None or self._plan["dev"] != net.device or net._net is None or net._net_sig != self._plan["net_sig"]:
            if self._plan is not None:
                _lib.load().linna_logprob_destroy(self._plan["handle"])
                self._plan = None
            self._build()
        return self._plan

    def _workspace(self, B, with_grad):
        key = (int(B), bool(with_grad))
        ws = self._ws.get(key)
        if ws is None:
            n = _lib.load().linna_logprob_ws_bytes(self._plan["handle"], int(B), 1 if with_grad else 0)
            ws = torch.empty(n // 4 + 4, dtype=torch.float32, device=self._plan["dev"])
            self._ws[key] = ws
        return ws

    def _to_device(self, x):
        p = self._ensure()
        if not torch.is_tensor(x):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        z = x.detach().to(device=p["dev"], dtype=torch.float32)
        one = z.dim() == 1
        z = z.view(1, -1) if one else z
        if z.shape[1] != p["nin"]:
            raise ValueError("expected %d parameters per walker, got %d" % (p["nin"], z.shape[1]))
        return z.contiguous(), one

    @staticmethod
    def _check_rows(z, p):
        """The kernels take ``z[B, >= nin]`` float32 on the plan's device with unit column stride (any row stride)."""
        if not (torch.is_tensor(z) and z.is_cuda and z.dtype == torch.float32 and z.dim() == 2):
            raise _lib.LinnaHipError("expected a 2-D float32 device tensor of walker positions")
        if z.device != p["dev"]:
            raise _lib.LinnaHipError("walker positions on %s, log-probability on %s" % (z.device, p["dev"]))
        if z.shape[1] < p["nin"] or (z.shape[0] > 1 and z.stride(1) != 1) or (z.shape[1] > 1 and z.stride(1) != 1):
            raise _lib.LinnaHipError("walker rows must hold >= %d contiguous columns (got shape %s, strides %s)"
                                     % (p["nin"], tuple(z.shape), tuple(z.stride())))

    # -------------------------------------------------------------- evaluation
    def evaluate(self, z, out=None, theta=None):
        """Device-to-device batch evaluation: ``z[B, nin]`` (cuda, fp32, row stride free) ->
        ``lnP[B]``.  No host synchronisation; graph-capturable."""
        p = self._ensure()
        self._check_rows(z, p)
        B = z.shape[0]
        if out is None:
            out = torch.empty(B, dtype=torch.float32, device=z.device)
        _lib.call("linna_logprob_eval", p["handle"], _lib.ptr(z) if z.is_contiguous() else C.c_void_p(z.data_ptr()),
                  z.stride(0), B, _lib.ptr(self._workspace(B, False)), _lib.ptr(out),
                  C.c_void_p(theta.data_ptr()) if theta is not None else None,
                  theta.stride(0) if theta is not None else 0, _lib.stream())
        return out

    def evaluate_with_grad(self, z, out=None, grad=None):
        """``(lnP[B], d lnP/d z [B, nin])`` -- what ``torch.autograd.grad(lnP, x)`` yields in
        HMCSampler.py:32, batched per walker."""
        p = self._ensure()
        self._check_rows(z, p)
        B = z.shape[0]
        if out is None:
            out = torch.empty(B, dtype=torch.float32, device=z.device)
        if grad is None:
            grad = torch.empty((B, p["nin"]), dtype=torch.float32, device=z.device)
        _lib.call("linna_logprob_grad", p["handle"], C.c_void_p(z.data_ptr()), z.stride(0), B,
                  _lib.ptr(self._workspace(B, True)), _lib.ptr(out), C.c_void_p(grad.data_ptr()), grad.stride(0),
                  _lib.stream())
        return out, grad

    @property
    def device_only(self):
        """True when every term of the log-probability is computed by the HIP pipeline: the Gaussian likelihood and no
        ``externalloglike``.  A user ``loglikelihoodfunc`` / ``externalloglike`` is host code (main.py:277-279,
        util.py:1003-1008): the emulator still runs on the GPU, the callbacks are applied per walker on the host."""
        return self.loglikelihoodfunc is gaussianlogliklihood and self.externalloglike is None

    def evaluate_any(self, z):
        """``lnP[B]`` (device float32) of ``z[B, >= nin]`` on the device, user callbacks included -- what the walker
        loops call; ``evaluate`` / ``linna_stretch_half_step`` are the launches behind it when ``device_only``."""
        p = self._ensure()
        if self.device_only:
            return self.evaluate(z)
        zc = z[:, :p["nin"]].contiguous()
        if self.loglikelihoodfunc is not gaussianlogliklihood:
            return self._generic(zc).to(z.device)
        theta = torch.empty_like(zc)
        like = self.evaluate(zc, theta=theta)
        th = theta.cpu().numpy()
        ext = np.array([np.float32(self.externalloglike(t)) for t in th], np.float32)
        like = like + torch.from_numpy(ext).to(like.device)
        return torch.where(torch.isnan(like), torch.full_like(like, -float("inf")), like)

    def __call__(self, x, returntorch=True, inputnumpy=True):
        z, one = self._to_device(x)
        if z.shape[0] == 0:                              # empty batch: nothing to launch
            like = torch.empty(0, dtype=torch.float32)
            return like if returntorch else like.numpy()
        like = self.evaluate_any(z).cpu()
        if one:
            like = like[0]
        return like if returntorch else like.numpy()

    def _generic(self, z):
        """User-supplied ``loglikelihoodfunc(m[1,nout], data, invcov)`` (main.py:277-278): the
        emulator still runs on the GPU, the callback is applied per walker on the host."""
        theta = self.transform(z.cpu(), inputnumpy=False, returnnumpy=False).reshape(z.shape[0], -1)
        m = self.y_invtransform_data(self.model.predict(theta, no_grad=True)).cpu()
        data = torch.as_tensor(_np32(self.data_new))
        invcov = torch.as_tensor(_np32(self.invcov_new))
        out = []
        for i in range(z.shape[0]):
            v = self.loglikelihoodfunc(m[i:i + 1], data, invcov) / self.T + lnprior(z[i].cpu())
            if self.externalloglike is not None:
                v = v + np.float32(self.externalloglike(theta[i].numpy()))
            out.append(float(v))
        out = torch.tensor(out, dtype=torch.float32)
        return torch.where(torch.isnan(out), torch.full_like(out, -float("inf")), out)


class Dlnp(object):
    """Gradient of ``Log_prob`` wrt the latent position -- the INTENDED semantics of
    util.py:1023-1035 (broken as shipped: SURVEY §8 a17)."""

    def __init__(self, data_new, invcov_new, model, y_invtransform_data, transform, temperature):
        self.log_prob = Log_prob(data_new, invcov_new, model, y_invtransform_data, transform, temperature)

    def __call__(self, x, lnP=None, returntorch=None, inputnumpy=None):
        z, one = self.log_prob._to_device(x)
        _, g = self.log_prob.evaluate_with_grad(z)
        g = g.cpu().numpy()
        return g[0] if one else g


class Ddlnp(object):
    """Hessian of ``Log_prob`` wrt the latent position -- the INTENDED semantics of util.py:1037-1051
    (same broken constructor as ``Dlnp`` in the reference).  The reference differentiates the
    autograd gradient row by row; here the rows are central differences of the HIP gradient,
    all 2*ndim displaced points evaluated in ONE batched ``linna_logprob_grad`` call."""

    def __init__(self, data_new, invcov_new, model, y_invtransform_data, transform, temperature, eps=1e-2):
        self.log_prob = Log_prob(data_new, invcov_new, model, y_invtransform_data, transform, temperature)
        self.eps = float(eps)

    def __call__(self, x):
        z, _ = self.log_prob._to_device(x)
        n = z.shape[1]
        e = self.eps * torch.eye(n, device=z.device, dtype=torch.float32)
        pts = torch.cat([z[0:1] + e, z[0:1] - e]).contiguous()
        _, g = self.log_prob.evaluate_with_grad(pts)
        g = g.double()
        hess = (g[:n] - g[n:]) / (2 * self.eps)
        return (0.5 * (hess + hess.T)).cpu().numpy()


# ------------------------------------------------------------------ training objects (util.py:383-400, 1055-1127)
class ArrayDataset(object):
    """util.py:383-400: float32 views of the training arrays."""

    def __init__(self, X, y):
        self.X = np.asarray(X).astype(np.float32)
        self.y = np.asarray(y).astype(np.float32)

    def __len__(self):
        return self.X.shape[0]

    def __getitem__(self, i):
        return self.X[i, :], self.y[i, :]


class Auxilleryfunc(object):
    """Constants of the chi^2-ratio loss (util.py:1055-1088): the inverse of the covariance in
    the network's normalised output space (built in fp64, stored fp32) and the normalised
    data vector.  The arithmetic itself runs in ``linna_chi2_ratio_loss_fwd_bwd``."""

    def __init__(self, data_in, cov_tensor, inv_cov_tensor, y_transform_data, y_inv_transform, device="cpu"):
        self.inv_cov_tensor = inv_cov_tensor
        self.transformed_cov = y_inv_transform.transform_cov(y_transform_data.transform_cov(cov_tensor))
        inv = torch.inverse(self.transformed_cov)
        # the reference uses inverse() as is; it is symmetric to ~1 ulp -- symmetrise in fp64 so
        # that the analytic gradient -2 C delta is exact for the matrix actually used
        self.inv_transformed_cov = (0.5 * (inv + inv.T)).type(torch.float32).detach()
        self.y_transform_data = y_transform_data
        self.y_inv_transform = y_inv_transform
        self.device = device
        self.data = data_in
        d = torch.as_tensor(_np32(data_in))
        self.data_in = torch.nan_to_num(y_inv_transform(y_transform_data(d)), nan=1e-30).detach().reshape(-1)

    def arrays(self):
        """(sigma, y_mean, y_std, data_norm, Cinv) as float32 numpy arrays."""
        return (_np32(self.y_transform_data.sigma), _np32(self.y_inv_transform.y_mean), _np32(self.y_inv_transform.y_std),
                _np32(self.data_in), _np32(self.inv_transformed_cov))


class Loss_fn(object):
    """Training loss: mean over the batch of chi2(target, pred)/max(chi2(target, data), nout/2)
    (util.py:1090-1116)."""

    def __init__(self, data_in, cov_tensor, inv_cov_tensor, y_transform_data, y_inv_transform, device="cpu"):
        self.auxileryfunction = Auxilleryfunc(data_in, cov_tensor, inv_cov_tensor, y_transform_data, y_inv_transform, device)


class Val_metric_fn(Loss_fn):
    """Validation metric [median loss, max |chi2_nnd/chi2_Md - 1|, median of the same] (util.py:1118-1127)."""


def median_absolute_deviation(y, median, dim):
    """util.py:1308-1313 (torch.median: lower middle element)."""
    return torch.abs(y - median).median(axis=dim).values


_TXT_CACHE = {}


def _loadtxt_cached(path):
    """``np.loadtxt`` of a sample file, parsed once per (path, size, mtime): every iteration of ``ml_sampler_core`` re-reads the
    ``*_samples_x.txt`` of ALL earlier iterations (util.py:1346-1373) -- ten parses of 10 000 x ndim text for a four-iteration
    run (bench.py `e2e`: train_NN.load_samples).  The array handed out is a copy."""
    st = os.stat(path)
    key = (os.path.abspath(path), st.st_size, st.st_mtime_ns)
    hit = _TXT_CACHE.get(key)
    if hit is None:
        if len(_TXT_CACHE) >= 64:
            _TXT_CACHE.clear()
        hit = _TXT_CACHE[key] = np.loadtxt(path)
    return hit.copy()


def _load_samples(outdir_list, usebest=False):
    """util.py:1342-1409: concatenate the samples of all iterations so far; ``usebest``: with the optimizer-seeded samples
    (``best_samples_*``, util.py:1235-1252) in front."""
    tx, ty, vx, vy = [], [], [], []
    for outdir in outdir_list:
        for lst, name, loader in ((tx, "train_samples_x.txt", _loadtxt_cached), (ty, "train_samples_y.npy", np.load),
                                  (vx, "val_samples_x.txt", _loadtxt_cached), (vy, "val_samples_y.npy", np.load)):
            a = loader(os.path.join(outdir, name))
            if len(a) > 1:
                lst.append(a)
    train_x, train_y, val_x, val_y = (np.concatenate(v) for v in (tx, ty, vx, vy))
    train_y_last = np.load(outdir_list[0] + "train_samples_y.npy")
    if len(train_y_last) == 0:
        train_y_last = train_y
    if usebest:                                                                            # util.py:1375-1408
        bx = [a for a in (np.loadtxt(o + "best_samples_x.txt") for o in outdir_list) if len(a) > 1]
        by = [a for a in (np.load(o + "best_samples_y.npy") for o in outdir_list) if len(a) > 1]
        try:
            bx, by = np.concatenate(bx), np.concatenate(by)
        except ValueError:
            bx, by = np.array(bx), np.array(by)
        if bx.ndim > 1:
            if train_x.ndim > 1:
                train_x, train_y = np.concatenate([bx, train_x]), np.concatenate([by, train_y])
            else:
                train_x, train_y, train_y_last = bx, by, by
        vbx = np.concatenate([np.loadtxt(o + "best_samples_x_val.txt") for o in outdir_list])
        vby = np.concatenate([np.load(o + "best_samples_y_val.npy") for o in outdir_list])
        if vbx.ndim > 1:
            if val_x.ndim > 1:
                val_x, val_y = np.concatenate([vbx, val_x]), np.concatenate([vby, val_y])
            else:
                val_x, val_y = vbx, vby
    return train_x, train_y, val_x, val_y, train_y_last


def train_nn(outdir, model, train_x, train_y, val_x, val_y, X_transform, y_transform, loss_fn, val_metric_fn, dev="cpu",
             verbose=False, retrain=True, pool=None, nocpu=False, size=0, rank=0, params=None, dist_group=None):
    """util.py:1272-1306: wrap the arrays and run ``Predictor.train``."""
    if not retrain and os.path.isfile(os.path.join(outdir, "best.pth.tar")):
        return None
    model = predictor_gpu.Predictor(train_x.shape[-1], train_y.shape[-1], X_transform=X_transform, y_transform=y_transform,
                                    device=dev, optim="automatic", model=model, scheduler=None, outdir=outdir)
    loader = predictor_gpu.BatchLoader(ArrayDataset(train_x, train_y), batch_size=params["batch_size"], shuffle=True,
                                       drop_last=True)
    val_loader = predictor_gpu.BatchLoader(ArrayDataset(val_x, val_y), batch_size=len(val_y), shuffle=False,
                                           drop_last=False)
    model.train_history = model.train(loader, params["num_epochs"], loss_fn, val_loader, val_metric_fn, initfrombest=True,
                                      pool=None, nocpu=nocpu, rank=rank, size=size, dist_group=dist_group,
                                      checkpoint_every=params.get("checkpoint_every", 1))
    return model


def train_NN(nnsampler, cov, inv_cov, sigma, outdir_in, outdir_list, data, dolog10index=None, ypositive=False, retrain=True,
             norder=2, temperature=None, docuda=False, pool=None, tsize=1, nnmodel_in=None, params=None, usebest=False,
             device=None, dist_group=None, rank=0):
    """Prepare statistics, transforms and loss, then train (util.py:1315-1472).  Same positional
    signature as the reference (``model_args.pkl`` holds the first 18 arguments, main.py:197).
    ``docuda`` is accepted for parity; training always runs on the GPU here."""
    if device is None:
        device = "cuda"
    sigma = np.asarray(sigma)
    y_transform_data = Y_transform_data(sigma, device="cpu")
    y_invtransform_data = Y_invtransform_data(sigma, device="cpu")
    if rank == 0:
        y_transform_data.pickle(os.path.join(outdir_in, "y_transform_data.pkl"))
        y_invtransform_data.pickle(os.path.join(outdir_in, "y_invtransform_data.pkl"))
    data_tensor = torch.from_numpy(np.asarray(data).astype(np.float32))
    with _lib.stage("train_NN.load_samples"):
        train_x, train_y, val_x, val_y, train_y_last = _load_samples(outdir_list, usebest)
    print(train_x.shape, train_y.shape, val_x.shape, val_y.shape)
    if ypositive:
        # util.py:1410-1431: positive data vectors are emulated in log space.  Clip to [1e-30, 1e10] (both ends are the
        # loss's mask sentinels, util.py:1072), drop rows that are 1e-30 throughout -- with the reference's own loop, which
        # deletes by the indices found BEFORE the first deletion (two or more such rows: the later deletions hit the rows
        # one further down, as there)
        train_y, val_y, train_y_last = np.array(train_y, np.float64), np.array(val_y, np.float64), np.array(train_y_last, np.float64)
        train_y[np.where(train_y > 1e10)] = 1e10
        train_y[np.where(train_y < 1e-30)] = 1e-30
        train_y_last[np.where(train_y_last < 1e-30)] = 1e-30
        val_y[np.where(val_y > 1e10)] = 1e10
        val_y[np.where(val_y < 1e-30)] = 1e-30
        for item in np.where(np.mean(train_y, axis=1) == 1e-30)[0]:
            train_y = np.delete(train_y, item, 0)
            train_x = np.delete(train_x, item, 0)
        for item in np.where(np.mean(train_y_last, axis=1) == 1e-30)[0]:
            train_y_last = np.delete(train_y_last, item, 0)
        for item in np.where(np.mean(val_y, axis=1) == 1e-30)[0]:
            val_y = np.delete(val_y, item, 0)
            val_x = np.delete(val_x, item, 0)
    else:
        # sentinel clipping, util.py:1433-1438
        train_y = np.clip(train_y, -1e5, 1e10)
        val_y = np.clip(val_y, -1e5, 1e8)
        train_y_last = np.clip(train_y_last, -1e5, 1e10)
    X1 = torch.tensor(train_x, dtype=torch.float32)
    if dolog10index is not None:
        for ind in dolog10index:
            X1[:, ind] = torch.log10(X1[:, ind])
    X_mean, X_std = X1.mean(axis=0), X1.std(axis=0)                                       # util.py:1440-1441
    X_transform = X_transform_class(X_mean, X_std, "cpu", dolog10index)
    if ypositive:
        # util.py:1444-1447: median / median absolute deviation of log(y / sigma) over ALL training rows (no floor on y_std)
        ys = torch.log(y_transform_data(torch.tensor(train_y, dtype=torch.float32)))
        y_mean = ys.median(axis=0).values
        y_std = median_absolute_deviation(ys, y_mean, 0)
    else:
        ys = y_transform_data(torch.tensor(train_y_last, dtype=torch.float32))
        y_mean = ys.median(axis=0).values                                                 # util.py:1449
        y_std = median_absolute_deviation(ys, y_mean, 0)
        y_std[y_std < 1e-10] = 1.0                                                        # util.py:1451
    y_transform = Y_transform_class(y_mean, y_std, "cpu", ypositive=bool(ypositive))
    y_inv_transform = Y_invtransform_class(y_mean, y_std, data_tensor, "cpu", ypositive=bool(ypositive))
    if rank == 0:
        X_transform.pickle(os.path.join(outdir_in, "X_transform.pkl"))
        y_transform.pickle(os.path.join(outdir_in, "y_transform.pkl"))
        y_inv_transform.pickle(os.path.join(outdir_in, "y_invtransform.pkl"))
    cov_t = torch.tensor(np.asarray(cov), dtype=torch.float64)
    icov_t = torch.tensor(np.asarray(inv_cov), dtype=torch.float64)
    loss_fn = Loss_fn(data_tensor, cov_t, icov_t, y_transform_data, y_inv_transform, "cpu")
    val_metric_fn = Val_metric_fn(data_tensor, cov_t, icov_t, y_transform_data, y_inv_transform, "cpu")
    nnmodel = nnmodel_in(len(train_x[0]), len(train_y[0]), None, docpu=False)
    if nnsampler is not None:
        nnsampler.model = nnmodel
    return train_nn(outdir_in, nnmodel, train_x, train_y, val_x, val_y, X_transform, y_transform, loss_fn, val_metric_fn,
                    dev=device, verbose=True, retrain=retrain, pool=pool, nocpu=True, size=tsize, rank=rank, params=params,
                    dist_group=dist_group)


# ------------------------------------------------------------------ training-point generation (util.py:736-897, 1167-1270)
class _FunctionWrapper(object):
    def __init__(self, f, args, kwargs):
        self.f, self.args, self.kwargs = f, ([] if args is None else args), ({} if kwargs is None else kwargs)

    def __call__(self, x):
        return self.f(x, *self.args, **self.kwargs)


def _apply_cuts(samples, omegab2cut):
    """util.py:804-811: optional cut on omega_b h^2 and up to two further parameter ranges."""
    if omegab2cut is None:
        return samples
    ombh2 = samples[:, omegab2cut[0]] * samples[:, omegab2cut[1]] ** 2
    keep = (ombh2 > omegab2cut[2]) & (ombh2 < omegab2cut[3])
    if len(omegab2cut) > 4:
        keep &= (samples[:, omegab2cut[4]] > omegab2cut[5]) & (samples[:, omegab2cut[4]] < omegab2cut[6])
    if len(omegab2cut) > 6:
        keep &= (samples[:, omegab2cut[7]] > omegab2cut[8]) & (samples[:, omegab2cut[7]] < omegab2cut[9])
    return samples[keep]


class NN_samplerv1(object):
    """Per-iteration helper of the reference (util.py:736-951): training-point designs and the
    MCMC launchers.  The user's ``theory`` callback is evaluated exactly as in the reference
    (``pool.map`` or ``map`` over ``(index, params)`` tuples)."""

    def __init__(self, outdir, prior_range):
        self.outdir, self.prior_range = outdir, prior_range
        self.seed = 123456
        self.model = None

    def generate_training_data(self, samples, model, pool=None, args=None, kwargs=None):
        m = _FunctionWrapper(model, args, kwargs)
        mapper = pool.map if pool is not None else map
        return np.array(list(mapper(m, samples)))

    def gensample_flat(self, Nsamples, omegab2cut=None):
        """util.py:775-814: ``pyDOE2.lhs(ndim, samples=n, criterion="center", iterations=5, random_state=seed)`` scaled to
        the prior box, grown by 1000 points until enough survive the cuts.  pyDOE2 (third party, absent here) is
        restated from its published algorithm -- ``_lhscentered``: cell centres ``(cut[i] + cut[i+1]) / 2`` of
        ``linspace(0, 1, n + 1)``, one unused ``rand(n, ndim)`` draw, then one ``permutation`` of the centres per
        column, all from ``RandomState(seed)`` (``iterations`` does not enter the centred design) -- and PINNED: with
        the reference's seed it reproduces the training and validation designs its own fixture holds
        (tests/test_data/.../train_samples_x.txt, val_samples_x.txt) bit for bit (tests/test_host_api.py)."""
        n_in, samples = int(Nsamples), np.zeros((0, len(self.prior_range)))
        ndim = len(self.prior_range)
        while len(samples) < Nsamples:
            rs = np.random.RandomState(self.seed)
            cut = np.linspace(0, 1, n_in + 1)
            rs.rand(n_in, ndim)                                         # drawn and dropped by pyDOE2's _lhscentered
            centre = (cut[:n_in] + cut[1:n_in + 1]) / 2
            s = np.zeros((n_in, ndim))
            for j in range(ndim):
                s[:, j] = rs.permutation(centre)
            s -= 0.5
            s *= 2
            shift_as = False
            for ind, prior in enumerate(self.prior_range):
                if ind == 1 and self.prior_range[1][1] < 1e-5:          # A_s sampled in log (util.py:795-803)
                    prior = np.log(prior)
                    shift_as = True
                scaled, mean = (prior[1] - prior[0]) / 2, (prior[1] + prior[0]) / 2
                s[:, ind] = s[:, ind] * scaled + mean
                if shift_as and ind == 1:
                    s[:, ind] = np.exp(s[:, ind])
            samples = _apply_cuts(s, omegab2cut)
            n_in += 1000
        return samples[:Nsamples]

    def gensample_chain_randomsample(self, Nsamples, chain_in, nsigma, omegab2cut=None):
        """util.py:864-897: uniform random draws (seed 123456) from the previous chain inside the prior box."""
        # (the reference filters column by column, one boolean-indexed copy of the chain per parameter: the same rows in the
        #  same order from ONE mask -- at 33 parameters and a million kept samples the copies were the largest host stage of an
        #  iteration that was not the user's theory code, bench.py `e2e`)
        chain = np.asarray(chain_in)
        if omegab2cut is not None:
            chain = _apply_cuts(np.array(chain, copy=True), omegab2cut)
        pr = np.asarray(self.prior_range, dtype=np.float64)
        chain = chain[np.all((chain > pr[None, :, 0]) & (chain < pr[None, :, 1]), axis=1)]
        np.random.seed(self.seed)
        return chain[np.random.randint(0, len(chain), int(Nsamples))]

    def emcee_sample(self, log_prob, ndim, nwalkers, init, pool, transform, ntimes=50, tautol=0.01, dlnp=None, ddlnp=None,
                     meanshift=0.1, stdshift=0.1, nk=1, max_n=1000000):
        from . import sampler
        x0 = init + 0.1 * np.random.randn(nwalkers, ndim)                 # util.py:915
        samp = sampler.HMCSampler(log_prob, dlnp, ddlnp, ndim, nwalkers, x0=x0, m=None, transform=transform)
        return samp.sample(pool, max_n, 0, 0, outdir=self.outdir, overwrite=False, ntimes=ntimes, method="emcee",
                           incremental=True, progress=False, tautol=tautol, meanshift=meanshift, stdshift=stdshift, nk=nk)

    def Zeus_sample(self, log_prob, ndim, nwalkers, init, pool, transform, ntimes=50, tautol=0.01, dlnp=None, ddlnp=None,
                    meanshift=0.1, stdshift=0.1, nk=1, max_n=1000000):
        from . import sampler
        x0 = init + 0.001 * np.random.randn(nwalkers, ndim)               # util.py:937
        samp = sampler.ZeusSampler(log_prob, ndim, nwalkers, x0=x0, transform=transform)
        return samp.sample(pool, max_n, outdir=self.outdir, overwrite=False, ntimes=ntimes, incremental=True, progress=False,
                           tautol=tautol, meanshift=meanshift, stdshift=stdshift, nk=nk)


def chi2_rows(d, invcov, device=None, chunk=8192):
    """``d_i^T invcov d_i`` for every row of ``d[n, nout]`` in FLOAT64 on the host, as the reference computes the two
    quantities this feeds (``logp_theory_data``, util.py:1506-1517: the importance weights of main.py:297-334;
    ``chisqcut_all``, util.py:1260-1270).  Real inverse covariances are ill-conditioned and the fp32 error of a
    quadratic form grows with the condition number, which would bias ``w = exp(logp - lp)`` directly; these are one-off
    post steps of n x nout^2 work, so nothing is gained by the GPU here (``chi2_rows_gpu`` is the fp32 device form)."""
    d = np.ascontiguousarray(np.atleast_2d(np.asarray(d, np.float64)))
    S = np.asarray(invcov, np.float64)
    out = np.empty(len(d))
    for lo in range(0, len(d), chunk):
        blk = d[lo:lo + chunk]
        out[lo:lo + len(blk)] = np.einsum("bi,bi->b", blk @ S, blk)
    return out


def chi2_rows_gpu(d, invcov, device=None, chunk=32768):
    """``d_i^T invcov d_i`` for every row of ``d[n, nout]`` on the GPU: the dense log-likelihood entry
    (``linna_gauss_loglike_dense``: MFMA GEMM d.S fused with the row-dot) with T = 1 and no prior term returns
    -chi2/2.  fp32 arithmetic on rows handed over in float32; returns float64 [n].  Well-conditioned problems only
    (relative error ~ 1e-6 x the condition number): the walker loop's own quantity, not the importance weights'."""
    d = np.ascontiguousarray(np.atleast_2d(np.asarray(d, np.float64)))
    n, nout = d.shape
    if n == 0:
        return np.zeros(0)
    dev = torch.device(device if device is not None else "cuda")
    if dev.type != "cuda" or not torch.cuda.is_available():
        raise _lib.LinnaHipError("chi2_rows_gpu runs on the GPU (no CPU fallback)")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    ld = _lib.ld4(nout)
    S = torch.as_tensor(np.pad(np.asarray(invcov, np.float32), ((0, 0), (0, ld - nout))), device=dev)
    out = np.empty(n)
    lib, ctx = _lib.load(), _lib.ctx(dev.index)
    for lo in range(0, n, chunk):
        blk = d[lo:lo + chunk]
        B = len(blk)
        D = torch.zeros((B, ld), dtype=torch.float32, device=dev)
        D[:, :nout].copy_(torch.as_tensor(blk.astype(np.float32)))
        scratch = torch.empty(B * lib.linna_gemm_dot_slots(B, nout) + 4, dtype=torch.float32, device=dev)
        res = torch.empty(B, dtype=torch.float32, device=dev)
        _lib.call("linna_gauss_loglike_dense", ctx, _lib.ptr(D), ld, B, nout, _lib.ptr(S), ld, _lib.ptr(D), ld, 0, 1.0,
                  _lib.ptr(scratch), _lib.ptr(res), _lib.stream())
        out[lo:lo + B] = -2.0 * res.double().cpu().numpy()
    return out


def chisqcut_all(data, invcov, chisqcut, fnamey, fnamex):
    """util.py:1260-1270: drop the training rows whose ``y^T invcov y`` reaches ``chisqcut`` (the reference measures
    the theory vector itself here, not its distance to ``data``; kept as is).  Float64 on the host, as the reference."""
    y, x = np.load(fnamey), np.loadtxt(fnamex)
    chisq = chi2_rows(y, invcov)
    np.save(fnamey, y[chisq < chisqcut])
    np.savetxt(fnamex, x[chisq < chisqcut])


def generate_training_point(theory, nnsampler, pool, outdir, ntrain, nval, data, invcov, chain=None, nsigma=1,
                            omegab2cut=None, options=0, negloglike=None, nbest_in=None, chisqcut=None):
    """util.py:1167-1258: design the training / validation parameters, evaluate the user's theory
    on them, store ``{train,val}_samples_{x.txt,y.npy}`` (skipping whatever already exists)."""
    if not (pool is None or pool.is_master()):
        return
    os.makedirs(outdir, exist_ok=True)

    def design(n):
        if chain is None:
            return nnsampler.gensample_flat(n, omegab2cut=omegab2cut)
        if options == 1:
            return nnsampler.gensample_chain_randomsample(n, chain, nsigma, omegab2cut=omegab2cut)
        raise NotImplementedError("options=0 needs the third-party sample_generator package (util.py:841)")

    for tag, n in (("train", ntrain), ("val", nval)):
        fx, fy = os.path.join(outdir, tag + "_samples_x.txt"), os.path.join(outdir, tag + "_samples_y.npy")
        if not os.path.isfile(fx):
            with _lib.stage("training_points.design"):
                pts = design(n)
            with _lib.stage("training_points.text_io"):
                np.savetxt(fx, pts)
        sub = os.path.join(outdir, tag + "/")
        os.makedirs(sub, exist_ok=True)
        if not os.path.isfile(fy):
            with _lib.stage("training_points.text_io"):
                x = np.loadtxt(fx)
            with _lib.stage("training_points.theory"):
                y = nnsampler.generate_training_data(zip(range(len(x)), x), theory, pool=pool, args=[sub])
            np.save(fy, y)
        if chisqcut is not None:
            chisqcut_all(data, invcov, chisqcut, fy, fx)
    if negloglike is not None:
        # util.py:1235-1252 (`nbest`): one Nelder-Mead fit of the true theory from the first training point, then nbest_in
        # (and nbest_in nval / ntrain validation) draws from N(best fit, inverse Hessian) and the theory at them.  The
        # reference takes the Hessian from numdifftools (third party, absent from its tree and from this image): here central
        # second differences with steps of 1e-4 (|x| + 1e-2) -- PARITY UNPINNED for that matrix; the draws are unseeded there too.
        from scipy.optimize import minimize
        from scipy.stats import multivariate_normal
        import tempfile
        fbx, fbv = os.path.join(outdir, "best_samples_x.txt"), os.path.join(outdir, "best_samples_x_val.txt")
        if not os.path.isfile(fbx):
            x0 = np.loadtxt(os.path.join(outdir, "train_samples_x.txt"))[0]
            best = minimize(negloglike, x0, method="Nelder-Mead", tol=1e-6).x
            widths = np.array([hi - lo for lo, hi in nnsampler.prior_range], np.float64)
            inv_hess = np.linalg.inv(makepositivedefinite(numerical_hessian(negloglike, best, scale=widths)))
            np.savetxt(fbx, multivariate_normal.rvs(mean=best, cov=inv_hess, size=nbest_in, random_state=None))
            np.savetxt(fbv, multivariate_normal.rvs(mean=best, cov=inv_hess, size=int(nbest_in / ntrain * nval), random_state=None))
        if not os.path.isfile(os.path.join(outdir, "best_samples_y.npy")):
            for fx, fy in ((fbx, "best_samples_y.npy"), (fbv, "best_samples_y_val.npy")):
                x = np.loadtxt(fx)
                with tempfile.TemporaryDirectory() as tmp:
                    np.save(outdir + fy, nnsampler.generate_training_data(zip(range(len(x)), x), theory, pool=pool, args=[tmp]))
        if chisqcut is not None:
            chisqcut_all(data, invcov, chisqcut, os.path.join(outdir, "best_samples_y.npy"), fbx)
            chisqcut_all(data, invcov, chisqcut, os.path.join(outdir, "best_samples_y_val.npy"), fbv)


def makepositivedefinite(cov, fcut=0.99):
    """util.py:38-48 (without its stray plot): eigenvalues below zero set to zero, those past the `fcut` point of the
    cumulative spectrum raised to the value there."""
    eigvals, eigvec = np.linalg.eigh(cov)
    eigvals, eigvec = eigvals[::-1].copy(), eigvec[:, ::-1]
    eigvals[eigvals < 0] = 0
    cumsum = np.cumsum(eigvals)
    cumsum = cumsum / np.max(cumsum)
    ind = np.argmin(np.abs(cumsum - fcut))
    eigvals[ind:] = eigvals[ind]
    return eigvec @ np.diag(eigvals) @ eigvec.T


def numerical_hessian(f, x, rel_step=None, check=True, scale=None):
    """Central second differences of a scalar function -- stands in for ``numdifftools.Hessian`` (util.py:1240; third party,
    absent from the reference tree and from this image: PARITY UNPINNED for this matrix).

    Steps ``h_i = rel_step s_i`` with ``s_i = scale_i`` (the caller's parameter scales: ``generate_training_point`` passes the
    prior widths) or ``|x_i| + 1e-2`` without them.  ``f`` is the chi^2 of an EXTERNAL theory code whose own noise (an
    iterative solver, an interpolation table; the Nelder-Mead fit before this call stops at 1e-6) is amplified by
    ``1 / h^2`` in a second difference, so one fixed step is wrong for somebody: with ``rel_step=None`` each diagonal element
    is taken at 1e-3, 2e-3, ... (doubling, up to 1.6e-2) until two consecutive steps agree within 10 %, keeping the larger of
    the two -- or, if none do, the step of the closest pair (a step-doubling consistency check: the idea of numdifftools'
    Richardson sequence without its extrapolation); the off-diagonals use the steps so chosen.
    ``LINNA_HESSIAN_STEP`` / ``rel_step`` fix the step.  ``check``: warn when the matrix is not positive definite before
    ``makepositivedefinite`` reshapes its spectrum (util.py:38-48) -- the draws around the best fit then follow the repair,
    not the likelihood."""
    x = np.asarray(x, np.float64)
    n = len(x)
    if rel_step is None and os.environ.get("LINNA_HESSIAN_STEP"):
        rel_step = float(os.environ["LINNA_HESSIAN_STEP"])
    base = np.abs(x) + 1e-2 if scale is None else np.asarray(scale, np.float64)
    f0 = f(x)

    def diag(i, h):
        e = np.zeros(n); e[i] = h
        return (f(x + e) - 2 * f0 + f(x - e)) / h ** 2
    if rel_step is not None:
        h = rel_step * base
        d = np.array([diag(i, h[i]) for i in range(n)])
    else:
        steps = [1e-3 * 2 ** k for k in range(5)]
        h, d = np.empty(n), np.empty(n)
        for i in range(n):
            vals = [diag(i, steps[0] * base[i])]
            best = None                                       # (difference, index of the larger step of the pair)
            for k in range(1, len(steps)):
                vals.append(diag(i, steps[k] * base[i]))
                diff = abs(vals[k] - vals[k - 1])
                if best is None or diff < best[0]:
                    best = (diff, k)
                if diff <= 0.1 * max(abs(vals[k]), abs(vals[k - 1]), 1e-300):
                    best = (diff, k)
                    break
            h[i], d[i] = steps[best[1]] * base[i], vals[best[1]]
    H = np.diag(d)
    for i in range(n):
        ei = np.zeros(n); ei[i] = h[i]
        for j in range(i):
            ej = np.zeros(n); ej[j] = h[j]
            H[i, j] = H[j, i] = (f(x + ei + ej) - f(x + ei - ej) - f(x - ei + ej) + f(x - ei - ej)) / (4 * h[i] * h[j])
    if check:
        w = np.linalg.eigvalsh(0.5 * (H + H.T))
        if not np.all(w > 0):
            import warnings
            warnings.warn("numerical_hessian: %d of %d eigenvalues are not positive (smallest %.3g, largest %.3g): the best-fit "
                          "point is not a minimum at this step size, or the theory is too noisy for second differences; "
                          "makepositivedefinite will reshape the spectrum (set LINNA_HESSIAN_STEP to change the step)"
                          % (int((w <= 0).sum()), n, w.min(), w.max()), RuntimeWarning)
    return H


class LogPrior(object):
    """util.py:1129-1157 (theta-space prior used by the importance-sampling post step)."""

    def __init__(self, prior):
        self.prior = prior

    def __call__(self, xlist):
        logp = 0
        for ind, x in enumerate(xlist):
            item = self.prior[ind]
            if item["dist"] == "flat" and (x < item["arg1"] or x > item["arg2"]):
                return -np.inf
            if item["dist"] == "gauss":
                logp += -0.5 * (x - item["arg1"]) ** 2 / item["arg2"] ** 2
        return logp


def logp_theory_data(samples, theory, data, invcov, logprior):
    """util.py:1506-1517: ``-chi2/2 + logprior`` of the importance-sampling post step (main.py:297-334), the
    ``(t - data)^T invcov (t - data)`` of all rows in float64 (``chi2_rows``)."""
    theory = np.asarray(theory, np.float64)
    data = np.asarray(data, np.float64)
    d = theory[:, :len(data)] - data[None, :]
    chisq = chi2_rows(d, invcov)
    return [-0.5 * c + logprior(s) for c, s in zip(chisq, samples)]


def read_chain_and_cut(chainname, nk, ntimes=20, walkercut=False, method="emcee", flat=False):
    from . import sampler
    return sampler.read_chain_and_cut(chainname, nk, ntimes, walkercut, method, flat)


def run_mcmc(nnsampler, outdir, method, ndim, nwalkers, init, log_prob, dlnp=None, ddlnp=None, pool=None, transform=None,
             ntimes=50, tautol=0.01, meanshift=0.1, stdshift=0.1, nk=2):
    """util.py:1474-1504 (the "hmc"/"nuts" branches of the reference are unreachable)."""
    kw = dict(ntimes=ntimes, tautol=tautol, transform=transform, dlnp=dlnp, ddlnp=ddlnp, meanshift=meanshift,
              stdshift=stdshift, nk=nk)
    if method == "emcee":
        return nnsampler.emcee_sample(log_prob, ndim, nwalkers, init, pool, **kw)
    if method == "zeus":
        return nnsampler.Zeus_sample(log_prob, ndim, nwalkers, init, pool, **kw)
    raise NotImplementedError(method)
