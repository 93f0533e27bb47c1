#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the LIVE reference.

Runs only in the build container (needs /root/reference).  It imports the reference
package (with inert stubs for absent third-party modules, see _ref_import.py), feeds it
deterministic synthetic weights / inputs (synth.py) and stores inputs + expected outputs
as small .npz files.  It also copies the DATA files of the reference's own test fixture
(tests/test_data/2dgaussian_Fulltconn/iter_0) -- checkpoints, transform pickles, sample
arrays -- which the reference's tests/test_main.py:47-51 reads.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz

No reference source text is copied anywhere.
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import synth  # noqa: E402
import _ref_import  # noqa: E402

rnn, rutil, rpred, rhmc = _ref_import.import_reference()
import torch  # noqa: E402
from torch import nn as tnn  # noqa: E402

torch.set_num_threads(4)
FIX_SRC = os.path.join(_ref_import.REFERENCE_ROOT, "tests/test_data/2dgaussian_Fulltconn/iter_0")
FIX_DST = os.path.join(HERE, "2dgaussian_Fulltconn/iter_0")
FIX_FILES = ["best.pth.tar", "last.pth.tar", "X_transform.pkl", "y_transform.pkl", "y_invtransform.pkl",
             "y_transform_data.pkl", "y_invtransform_data.pkl", "train_samples_x.txt",
             "train_samples_y.npy", "val_samples_x.txt", "val_samples_y.npy", "lr.npy"]


class TorchMLP(tnn.Module):
    """Plain ReLU MLP (BASELINE configs 2/5); not a reference class, built here so that
    the reference's Predictor / Log_prob can drive it through the nnmodel_in signature."""

    def __init__(self, in_size, out_size, linearmodel=None, docpu=False, width=512, depth=4):
        super().__init__()
        k = in_size
        self.depth = depth
        for i in range(depth):
            setattr(self, "layer%d" % (i + 1), tnn.Linear(k, width))
            k = width
        setattr(self, "layer%d" % (depth + 1), tnn.Linear(k, out_size))

    def forward(self, s):
        for i in range(self.depth):
            s = torch.relu(getattr(self, "layer%d" % (i + 1))(s))
        return getattr(self, "layer%d" % (self.depth + 1))(s)


def build_model(kind, nin, nout, seed, **kw):
    cls = {"ChtoModelv2": rnn.ChtoModelv2, "ChtoModelsimple": rnn.ChtoModelsimple,
           "ChtoModelv2_linear": rnn.ChtoModelv2_linear, "MLP": TorchMLP}[kind]
    model = cls(nin, nout, None, **kw) if kind == "MLP" else cls(nin, nout, None)
    w = synth.weights(kind, nin, nout, seed, **kw)
    sd = {k: torch.from_numpy(v.copy()) for k, v in w.items()}
    model.load_state_dict(sd, strict=True)
    return model


def t32(a):
    return torch.from_numpy(np.asarray(a, np.float32))


# ------------------------------------------------------------------ serving cases
SERVING = [
    # name, kind, nin, nout, seed, dense, n, dolog10, ypositive, extra kw
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
TEMPS = [1.0, 4.0, 16.0]


def make_logprob(kind, nin, nout, seed, dense, dolog10, ypositive, kw, temperature, nograd=True):
    data, cov, priors = synth.gaussian_problem(nin, nout, seed, dense=dense)
    if dolog10 is not None:
        for i in dolog10:
            priors[i] = {"param": "p%d" % i, "dist": "flat", "arg1": 0.1, "arg2": 2.0}
    if ypositive:
        data = np.abs(data) + 0.5
    X_mean, X_std, y_mean, y_std = synth.transform_constants(nin, nout, seed)
    if ypositive:
        y_std = (0.1 * y_std).astype(np.float32)
    sigma = np.sqrt(np.diag(cov))
    invcov = np.linalg.inv(cov)
    model = build_model(kind, nin, nout, seed, **kw)
    Xt = rutil.X_transform_class(t32(X_mean), t32(X_std), "cpu", dolog10)
    Yt = rutil.Y_transform_class(t32(y_mean), t32(y_std), "cpu", ypositive=ypositive)
    pred = rpred.Predictor(nin, nout, model=model, X_transform=Xt, y_transform=Yt, device="cpu")
    yinv = rutil.Y_invtransform_data(sigma, "cpu")
    transform = rutil.Transform(priors)
    lp = rutil.Log_prob(t32(data), t32(invcov), pred, yinv, transform, temperature,
                        rutil.gaussianlogliklihood, nograd=nograd)
    return lp, pred, yinv, transform, dict(data=data, cov=cov, invcov=invcov, sigma=sigma)


def gen_serving(out):
    for name, kind, nin, nout, seed, dense, n, dolog10, ypos, kw in SERVING:
        z = synth.latent_points(n, nin, seed)
        lp, pred, yinv, transform, prob = make_logprob(kind, nin, nout, seed, dense, dolog10, ypos, kw, 1.0)
        theta = np.stack([np.atleast_1d(transform(zi)) for zi in z]).astype(np.float32)
        with torch.no_grad():
            m = np.stack([yinv(pred.predict(t32(th))[None, :])[0].numpy() for th in theta])
            # batched predict (predictor_gpu.py:461 accepts [B, nin])
            mb = yinv(pred.predict(t32(theta))).numpy()
        assert np.allclose(m, mb, rtol=2e-4, atol=2e-5), name
        ll = np.zeros((n, len(TEMPS)), np.float32)
        for j, T in enumerate(TEMPS):
            lpT = make_logprob(kind, nin, nout, seed, dense, dolog10, ypos, kw, T)[0]
            ll[:, j] = [float(lpT(zi)) for zi in z]
        # gradient of lnP wrt z (autograd through Log_prob(nograd=False), HMCSampler.py:32)
        lpg = make_logprob(kind, nin, nout, seed, dense, dolog10, ypos, kw, 1.0, nograd=False)[0]
        grads = np.zeros((n, nin), np.float32)
        for i, zi in enumerate(z):
            x = t32(zi).clone().requires_grad_()
            val = lpg(x, inputnumpy=False)
            grads[i] = torch.autograd.grad(val, x)[0].numpy()
        out[name] = dict(z=z, theta=theta, m=m.astype(np.float32), loglike=ll, grad=grads,
                         temps=np.array(TEMPS, np.float32))
        print("serving", name, "lnP[0..2] =", ll[:3, 0], flush=True)


# ------------------------------------------------------------------ fixture (2-D)
def _fixture_model():
    """The reference's ``retrieve_model`` (util.py:611-639) on its own fixture directory, with the FILES read by loaders
    that execute nothing: the checkpoint through ``torch.load(weights_only=True)``, the three transform pickles through
    this package's closed allow-list unpickler (their tensors move into the reference's transform classes)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from linna_amd import util as putil, nnutils as pnnutils
    src = FIX_SRC + "/"

    def safe(name):
        with open(src + name, "rb") as f:
            return putil.CPU_Unpickler(f).load()
    xt, yt, yid = safe("X_transform.pkl"), safe("y_transform.pkl"), safe("y_invtransform_data.pkl")
    X_transform = rutil.X_transform_class(xt.X_mean, xt.X_std, "cpu", xt.dolog10index)
    y_transform = rutil.Y_transform_class(yt.y_mean, yt.y_std, "cpu", ypositive=yt.ypositive)
    yinv = rutil.Y_invtransform_data(yid.sigma.detach().numpy(), "cpu")
    net = rnn.ChtoModelv2(2, 2, None)
    net.load_state_dict(pnnutils.read_checkpoint(src + "best.pth.tar")["state_dict"])
    model = rpred.Predictor(2, 2, X_transform=X_transform, y_transform=y_transform, device="cpu", outdir=src, model=net)
    return model, yinv


def gen_fixture(out):
    os.makedirs(FIX_DST, exist_ok=True)
    for f in FIX_FILES:
        shutil.copyfile(os.path.join(FIX_SRC, f), os.path.join(FIX_DST, f))
    model, yinv = _fixture_model()
    priors = [{"param": "test_%d" % i, "dist": "flat", "arg1": -2.0, "arg2": 2.0} for i in range(2)]
    cov = np.diag([0.5, 0.2])
    data = np.array([0.1, 1.0])
    transform = rutil.Transform(priors)
    lp = rutil.Log_prob(t32(data), t32(np.linalg.inv(cov)), model, yinv, transform, 1.0,
                        rutil.gaussianlogliklihood, nograd=True)
    z = np.concatenate([np.array([[0, 0], [0.3, -0.2], [1, 1], [-1.5, 0.7]], np.float32),
                        synth.latent_points(28, 2, 5)])
    vals = np.array([float(lp(zi)) for zi in z], np.float32)
    model.MKLDNN = True          # main.py:266-268 branch
    vals_mkl = np.array([float(lp(zi)) for zi in z], np.float32)
    assert np.allclose(vals, vals_mkl, rtol=1e-5, atol=1e-6)
    out["fixture2d"] = dict(z=z, loglike=vals,
                            X_mean=model.X_transform.X_mean.numpy(), X_std=model.X_transform.X_std.numpy(),
                            y_mean=model.y_transform.y_mean.numpy(), y_std=model.y_transform.y_std.numpy(),
                            sigma=yinv.sigma.detach().numpy())
    print("fixture2d", vals[:4], flush=True)


# ------------------------------------------------------------------ training
TRAIN = [
    ("train_v2_5_3", "ChtoModelv2", 5, 3, 201, 40, {}, True),
    ("train_mlp_7_5", "MLP", 7, 5, 202, 40, {"width": 48, "depth": 3}, True),
    ("train_v2_33_33", "ChtoModelv2", 33, 33, 203, 100, {}, False),
    ("train_v2_12_40", "ChtoModelv2", 12, 40, 204, 50, {}, False),
    ("train_v2_26_457", "ChtoModelv2", 26, 457, 205, 64, {}, False),      # BASELINE config 3 shape
    ("train_v2lin_5_3", "ChtoModelv2_linear", 5, 3, 206, 40, {}, True),    # nn.py:136-198: the input skip trains too
    ("train_simple_6_4", "ChtoModelsimple", 6, 4, 207, 40, {}, True),      # nn.py:300-374
]


def training_problem(nin, nout, seed, B):
    rs = np.random.RandomState(seed + 31)
    data, cov, _ = synth.gaussian_problem(nin, nout, seed, dense=True, cond=1e2)
    sigma = np.sqrt(np.diag(cov))
    X_mean, X_std, y_mean, y_std = synth.transform_constants(nin, nout, seed)
    X = (X_mean[None, :] + X_std[None, :] * rs.standard_normal((3, B, nin))).astype(np.float32)
    Y = (data[None, None, :] + 3 * sigma[None, None, :] * rs.standard_normal((3, B, nout))).astype(np.float32)
    Y[0, 1, 0] = 1e10        # sentinel masks (util.py:1072)
    Y[1, 2, nout - 1] = 1e-30
    return data, cov, sigma, X_mean, X_std, y_mean, y_std, X, Y


def gen_training(out):
    only = [n for n in os.environ.get("GOLDEN_TRAIN_ONLY", "").split(",") if n]     # (add a case without rewriting the others)
    for name, kind, nin, nout, seed, B, kw, full in TRAIN:
        if only and name not in only:
            continue
        data, cov, sigma, X_mean, X_std, y_mean, y_std, X, Y = training_problem(nin, nout, seed, B)
        model = build_model(kind, nin, nout, seed, **kw)
        ytd = rutil.Y_transform_data(sigma, device="cpu")
        data_t = t32(data)
        yinv = rutil.Y_invtransform_class(t32(y_mean), t32(y_std), data_t, "cpu", ypositive=False)
        cov_t = torch.tensor(cov, dtype=torch.float64)
        icov_t = torch.tensor(np.linalg.inv(cov), dtype=torch.float64)
        loss_fn = rutil.Loss_fn(data_t, cov_t, icov_t, ytd, yinv, "cpu")
        val_fn = rutil.Val_metric_fn(data_t, cov_t, icov_t, ytd, yinv, "cpu")
        Xt = rutil.X_transform_class(t32(X_mean), t32(X_std), "cpu", None)
        lr = 1e-3
        opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=1e-4)
        rec = dict(icov_norm=loss_fn.auxileryfunction.inv_transformed_cov.numpy(),
                   data_norm=loss_fn.auxileryfunction.data_in.numpy(), lr=np.float32(lr))
        losses = []
        for s in range(3):
            opt.zero_grad()
            pred = model(Xt(t32(X[s])))
            pred.retain_grad()
            loss = loss_fn(pred, t32(Y[s]))
            loss.backward()
            if s == 0:
                rec["pred0"] = pred.detach().numpy().copy()
                rec["dpred0"] = pred.grad.numpy().copy()
                rec["val0"] = val_fn(pred.detach(), t32(Y[s])).numpy()
                l_b, cMd, cnnd = loss_fn.auxileryfunction(pred.detach(), t32(Y[s]))
                rec["loss_rows0"] = l_b.numpy()
                rec["chisqMd0"] = cMd.numpy()
                rec["chisqnnd0"] = cnnd.numpy()
                for k, p in model.named_parameters():
                    g = p.grad.numpy()
                    rec["grad0/" + k] = g.copy() if full else synth.tensor_digest(g)
            opt.step()
            losses.append(loss.item())
            for k, p in model.named_parameters():
                v = p.detach().numpy()
                rec["param%d/%s" % (s + 1, k)] = v.copy() if full else synth.tensor_digest(v)
        rec["losses"] = np.array(losses, np.float32)
        out[name] = rec
        print("training", name, losses, flush=True)


# ------------------------------------------------------------------ full train_NN run
def gen_train_nn(out):
    nin, nout, seed = 5, 3, 301
    rs = np.random.RandomState(seed)
    data, cov, _ = synth.gaussian_problem(nin, nout, seed, dense=True, cond=1e2)
    sigma = np.sqrt(np.diag(cov))
    A = rs.standard_normal((nout, nin)) * 0.3
    def theory(x):
        return data[None, :] + np.tanh(x @ A.T) * 3 * sigma[None, :]
    ntrain, nval, batch, nep = 200, 50, 50, 6
    train_x = rs.uniform(-1, 1, (ntrain, nin)); val_x = rs.uniform(-1, 1, (nval, nin))
    train_y = theory(train_x); val_y = theory(val_x)
    w0 = synth.weights("ChtoModelv2", nin, nout, seed)

    def factory(in_size, out_size, linearmodel, docpu=False):
        m = rnn.ChtoModelv2(in_size, out_size, linearmodel, docpu=docpu)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w0.items()})
        return m

    tmp = tempfile.mkdtemp(prefix="linna_golden_") + "/"
    np.savetxt(tmp + "train_samples_x.txt", train_x); np.save(tmp + "train_samples_y.npy", train_y)
    np.savetxt(tmp + "val_samples_x.txt", val_x); np.save(tmp + "val_samples_y.npy", val_y)
    np.save(tmp + "lr.npy", 2e-3)

    class _S(object):
        pass
    captured = {}
    orig_train = rpred.Predictor.train

    def spy(self, *a, **k):
        r = orig_train(self, *a, **k)
        captured["ret"] = r
        captured["state"] = {kk: vv.detach().numpy().copy() for kk, vv in self.model.state_dict().items()}
        return r
    rpred.Predictor.train = spy
    try:
        rutil.train_NN(_S(), cov, np.linalg.inv(cov), sigma, tmp, [tmp], data, None, False, True, 2, 1.0,
                       False, None, 1, factory, {"num_epochs": nep, "batch_size": batch}, False)
    finally:
        rpred.Predictor.train = orig_train
    train_losses, val_metrics = captured["ret"]
    import pickle
    with open(tmp + "X_transform.pkl", "rb") as f:
        Xt = rutil.CPU_Unpickler(f).load()
    with open(tmp + "y_transform.pkl", "rb") as f:
        Yt = rutil.CPU_Unpickler(f).load()
    best = torch.load(tmp + "best.pth.tar", map_location="cpu", weights_only=False)
    rec = dict(train_x=train_x, train_y=train_y, val_x=val_x, val_y=val_y, data=data, cov=cov,
               lr=np.float64(2e-3), num_epochs=np.int64(nep), batch_size=np.int64(batch),
               X_mean=Xt.X_mean.numpy(), X_std=Xt.X_std.numpy(), y_mean=Yt.y_mean.numpy(), y_std=Yt.y_std.numpy(),
               train_losses=np.asarray(train_losses, np.float64), val_metrics=np.asarray(val_metrics, np.float64),
               best_epoch=np.int64(best["epoch"]))
    for k, v in captured["state"].items():
        rec["final/" + k] = v
    for k, v in best["state_dict"].items():
        rec["best/" + k] = v.numpy()
    out["train_nn_run"] = rec
    shutil.rmtree(tmp, ignore_errors=True)
    print("train_NN run: losses", np.asarray(train_losses)[:4], "val", np.asarray(val_metrics)[:2], flush=True)


def _run_reference_train_nn(tmp, cov, sigma, data, w0, nep, batch, lr, ypositive=False, usebest=False):
    """The live reference's train_NN on the sample files under `tmp`; returns what Predictor.train saw and produced."""
    def factory(in_size, out_size, linearmodel, docpu=False):
        m = rnn.ChtoModelv2(in_size, out_size, linearmodel, docpu=docpu)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w0.items()})
        return m
    np.save(tmp + "lr.npy", lr)

    class _S(object):
        pass
    captured = {}
    orig_train = rpred.Predictor.train

    def spy(self, dataset, num_epochs, loss_fn, val_dataset, val_metric_fn, *a, **k):
        captured["icov_norm"] = loss_fn.auxileryfunction.inv_transformed_cov.numpy().copy()
        captured["data_norm"] = loss_fn.auxileryfunction.data_in.numpy().copy()
        captured["ntrain"], captured["nval"] = len(dataset.dataset), len(val_dataset.dataset)
        r = orig_train(self, dataset, num_epochs, loss_fn, val_dataset, val_metric_fn, *a, **k)
        captured["ret"] = r
        captured["state"] = {kk: vv.detach().numpy().copy() for kk, vv in self.model.state_dict().items()}
        return r
    rpred.Predictor.train = spy
    try:
        rutil.train_NN(_S(), cov, np.linalg.inv(cov), sigma, tmp, [tmp], data, None, ypositive, True, 2, 1.0,
                       False, None, 1, factory, {"num_epochs": nep, "batch_size": batch}, usebest)
    finally:
        rpred.Predictor.train = orig_train
    with open(tmp + "X_transform.pkl", "rb") as f:
        Xt = rutil.CPU_Unpickler(f).load()
    with open(tmp + "y_transform.pkl", "rb") as f:
        Yt = rutil.CPU_Unpickler(f).load()
    best = torch.load(tmp + "best.pth.tar", map_location="cpu", weights_only=False)
    train_losses, val_metrics = captured["ret"]
    rec = dict(X_mean=Xt.X_mean.numpy(), X_std=Xt.X_std.numpy(), y_mean=Yt.y_mean.numpy(), y_std=Yt.y_std.numpy(),
               icov_norm=captured["icov_norm"], data_norm=captured["data_norm"], ntrain=np.int64(captured["ntrain"]),
               nval=np.int64(captured["nval"]), train_losses=np.asarray(train_losses, np.float64),
               val_metrics=np.asarray(val_metrics, np.float64), best_epoch=np.int64(best["epoch"]),
               lr=np.float64(lr), num_epochs=np.int64(nep), batch_size=np.int64(batch))
    for k, v in captured["state"].items():
        rec["final/" + k] = v
    return rec


def gen_train_nn_ypos(out):
    """`ypositive=True` (util.py:1410-1431, 1444-1447, 567-586): a positive data vector emulated in log space.  Targets
    with both sentinels (an entry above 1e10, entries at / below 0), one training row and one validation row that are
    non-positive throughout (deleted by the reference before the statistics)."""
    nin, nout, seed = 5, 4, 311
    rs = np.random.RandomState(seed)
    _, cov, _ = synth.gaussian_problem(nin, nout, seed, dense=True, cond=1e2)
    data = rs.uniform(5.0, 20.0, nout)                  # (the reference's log(1 + C / data^2) of the sigma-scaled covariance, util.py:579-583,
    sigma = np.sqrt(np.diag(cov))                       #  stays positive definite when its argument is small)
    A = rs.standard_normal((nout, nin)) * 0.3
    def theory(x):
        return data[None, :] * np.exp(0.5 * np.tanh(x @ A.T))
    ntrain, nval, batch, nep = 201, 51, 50, 4
    train_x = rs.uniform(-1, 1, (ntrain, nin)); val_x = rs.uniform(-1, 1, (nval, nin))
    train_y = theory(train_x); val_y = theory(val_x)
    train_y[3, 1] = 5e10; train_y[7, 0] = 0.0; train_y[11, 2] = -1.0       # clipped to the mask sentinels 1e10 / 1e-30
    train_y[20, :] = -3.0                                                   # a row that is 1e-30 throughout after clipping: dropped
    val_y[5, 3] = 0.0; val_y[9, :] = 0.0
    w0 = synth.weights("ChtoModelv2", nin, nout, seed)
    tmp = tempfile.mkdtemp(prefix="linna_golden_") + "/"
    np.savetxt(tmp + "train_samples_x.txt", train_x); np.save(tmp + "train_samples_y.npy", train_y)
    np.savetxt(tmp + "val_samples_x.txt", val_x); np.save(tmp + "val_samples_y.npy", val_y)
    rec = _run_reference_train_nn(tmp, cov, sigma, data, w0, nep, batch, 2e-3, ypositive=True)
    rec.update(train_x=train_x, train_y=train_y, val_x=val_x, val_y=val_y, data=data, cov=cov)
    out["train_nn_ypos"] = rec
    shutil.rmtree(tmp, ignore_errors=True)
    print("train_NN ypositive: rows", rec["ntrain"], rec["nval"], "losses", rec["train_losses"][:4], "val", rec["val_metrics"][:2], flush=True)


def gen_train_nn_usebest(out):
    """`usebest=True` (util.py:1375-1409): the optimizer-seeded samples of `nbest` in front of the designed ones."""
    nin, nout, seed = 5, 3, 321
    rs = np.random.RandomState(seed)
    data, cov, _ = synth.gaussian_problem(nin, nout, seed, dense=True, cond=1e2)
    sigma = np.sqrt(np.diag(cov))
    A = rs.standard_normal((nout, nin)) * 0.3
    def theory(x):
        return data[None, :] + np.tanh(x @ A.T) * 3 * sigma[None, :]
    ntrain, nval, nbest, batch, nep = 150, 40, 50, 50, 3
    xs = {"train": rs.uniform(-1, 1, (ntrain, nin)), "val": rs.uniform(-1, 1, (nval, nin)),
          "best": 0.1 * rs.standard_normal((nbest, nin)), "best_val": 0.1 * rs.standard_normal((int(nbest / ntrain * nval), nin))}
    w0 = synth.weights("ChtoModelv2", nin, nout, seed)
    tmp = tempfile.mkdtemp(prefix="linna_golden_") + "/"
    np.savetxt(tmp + "train_samples_x.txt", xs["train"]); np.save(tmp + "train_samples_y.npy", theory(xs["train"]))
    np.savetxt(tmp + "val_samples_x.txt", xs["val"]); np.save(tmp + "val_samples_y.npy", theory(xs["val"]))
    np.savetxt(tmp + "best_samples_x.txt", xs["best"]); np.save(tmp + "best_samples_y.npy", theory(xs["best"]))
    np.savetxt(tmp + "best_samples_x_val.txt", xs["best_val"]); np.save(tmp + "best_samples_y_val.npy", theory(xs["best_val"]))
    rec = _run_reference_train_nn(tmp, cov, sigma, data, w0, nep, batch, 2e-3, usebest=True)
    rec.update(data=data, cov=cov, **{"x_" + k: v for k, v in xs.items()}, **{"y_" + k: theory(v) for k, v in xs.items()})
    out["train_nn_usebest"] = rec
    shutil.rmtree(tmp, ignore_errors=True)
    print("train_NN usebest: rows", rec["ntrain"], rec["nval"], "losses", rec["train_losses"][:4], flush=True)


# ------------------------------------------------------------------ DataLoader order
def gen_loader_order(out):
    from torch.utils.data import DataLoader
    n, batch = 200, 50
    X = np.arange(n, dtype=np.float32)[:, None] * np.ones((1, 2), np.float32)
    ds = rutil.ArrayDataset(X, X)
    torch.manual_seed(1234)                           # predictor_gpu.py:221
    loader = DataLoader(ds, batch_size=batch, shuffle=True, drop_last=True, num_workers=0)  # util.py:1285
    order = []
    for ep in range(3):
        order.append(np.concatenate([xb[:, 0].numpy().astype(np.int64) for xb, _ in loader]))
    out["loader_order"] = dict(order=np.stack(order), n=np.int64(n), batch=np.int64(batch))


# ------------------------------------------------------------------ EarlyStopping traces
def gen_early_stopping(out):
    rs = np.random.RandomState(401)
    seqs = {}
    n = 1500
    t = np.arange(n)
    seqs["improve_then_plateau"] = (np.exp(-t / 120.0) + 0.05 + 0.002 * rs.standard_normal(n),
                                    np.exp(-t / 100.0) + 0.02 + 0.002 * rs.standard_normal(n))
    seqs["overfit"] = (0.2 + 0.3 * np.exp(-t / 50.0) + t * 2e-4 + 0.003 * rs.standard_normal(n),
                       0.5 * np.exp(-t / 200.0) + 0.003 * rs.standard_normal(n) + 0.01)
    seqs["noisy_flat"] = (1.0 + 0.05 * rs.standard_normal(n), 1.0 + 0.05 * rs.standard_normal(n))
    v = 1.0 / (1 + t / 30.0) + 0.01 * rs.standard_normal(n); v[700] = np.nan
    seqs["with_nan"] = (v, 1.0 / (1 + t / 25.0))
    rec = {}
    for name, (val, trn) in seqs.items():
        es = rpred.EarlyStopping(patience=500)
        codes = []
        for a, b in zip(val, trn):
            c = es.step(float(a), torch.tensor(float(b)))
            codes.append(int(c))
            if c == 2:
                break
        rec[name + "/val"] = np.asarray(val, np.float64)
        rec[name + "/train"] = np.asarray(trn, np.float64)
        rec[name + "/codes"] = np.asarray(codes, np.int64)
        print("early stopping", name, "len", len(codes), "codes used", sorted(set(codes)), flush=True)
    # a short-patience trace exercises codes 1 and 2 quickly
    es = rpred.EarlyStopping(patience=40, nqueue=20)
    val, trn = seqs["overfit"]
    codes = []
    for a, b in zip(val, trn):
        c = es.step(float(a), torch.tensor(float(b)))
        codes.append(int(c))
        if c == 2:
            break
    rec["overfit_p40/codes"] = np.asarray(codes, np.int64)
    out["early_stopping"] = rec


# ------------------------------------------------------------------ HMC trace
def gen_hmc(out):
    name, kind, nin, nout, seed, dense, n, dolog10, ypos, kw = SERVING[5]   # simple_6_4
    lp = make_logprob(kind, nin, nout, seed, dense, dolog10, ypos, kw, 1.0, nograd=False)[0]
    num_samps, num_steps, step = 40, 5, 0.02
    x0 = torch.zeros(nin)
    mass = torch.ones(nin)
    # replay the sampler's own draw order: torch.randn per sample (:26), np.random.uniform (:59)
    torch.manual_seed(77)
    momenta = np.stack([torch.randn(nin).numpy() for _ in range(num_samps)])
    np.random.seed(78)
    uniforms = np.array([np.random.uniform() for _ in range(num_samps)])
    torch.manual_seed(77)
    np.random.seed(78)
    s = rhmc.HMCSampler(lambda x: lp(x, inputnumpy=False), x0, mass)
    import linna.HMCSampler as mod
    mod.tqdm = lambda it: it
    chain = s.sample(num_samps, num_steps, step)
    out["hmc_trace"] = dict(momenta=momenta.astype(np.float32), uniforms=uniforms,
                            x=np.stack([c["x"] for c in chain]).astype(np.float32),
                            lnP=np.array([float(c["lnP"]) for c in chain], np.float32),
                            accepted=np.array([c["accepted"] for c in chain]),
                            num_steps=np.int64(num_steps), step_size=np.float64(step),
                            case=np.array(name))
    print("hmc accepted", int(sum(c["accepted"] for c in chain)), "of", num_samps, flush=True)


def callback_student_t(m, data, invcov):
    """A user likelihood in the reference's calling convention (util.py:953-955 is the default of this shape)."""
    d = m - data
    chi2 = (d @ invcov @ d.T)[0][0]
    return -0.5 * 5.0 * torch.log1p(chi2 / 4.0)


def callback_external(theta):
    return -0.25 * float(np.sum(np.asarray(theta) ** 2))


def gen_callbacks(out):
    """The callback surface of ``Log_prob`` (util.py:990-1021) in the LIVE reference, walker by walker: a user
    ``loglikelihoodfunc`` (Student-t) with an ``externalloglike`` of the theta-space parameters at T = 4, and the Gaussian
    default with the same ``externalloglike``."""
    name, kind, nin, nout, seed, dense, n, dolog10, ypos, kw = SERVING[5]   # simple_6_4
    lp, pred, yinv, transform, prob = make_logprob(kind, nin, nout, seed, dense, dolog10, ypos, kw, 4.0)
    z = (np.random.RandomState(3).standard_normal((40, nin)) * 0.5).astype(np.float32)
    lps = rutil.Log_prob(lp.data_new, lp.invcov_new, pred, yinv, transform, 4.0, callback_student_t, nograd=True,
                         externalloglike=callback_external)
    lpg = rutil.Log_prob(lp.data_new, lp.invcov_new, pred, yinv, transform, 4.0, rutil.gaussianlogliklihood, nograd=True,
                         externalloglike=callback_external)
    out["callbacks"] = dict(case=np.array(name), z=z,
                            student=np.array([float(lps(zi, returntorch=False)) for zi in z]),
                            gauss_ext=np.array([float(lpg(zi, returntorch=False)) for zi in z]))
    print("callbacks", out["callbacks"]["student"][:3], out["callbacks"]["gauss_ext"][:3], flush=True)


def gen_hmc_move(out):
    """The per-walker HMC move the reference WROTE but cannot reach through emcee (sampler.py:59-98 ``_hmc_wrapper``,
    :311-320 ``_hmc_matrix``): its integrator called directly, walker by walker, with the gradient of the
    reference's own ``Log_prob`` (autograd) -- proposal q and kinetic-energy factor for given momenta, plus the
    log-probabilities the Metropolis test of ``HamiltonianMove.propose`` (:141-143) would use."""
    import linna.sampler as rsamp
    name, kind, nin, nout, seed, dense, n, dolog10, ypos, kw = SERVING[5]   # simple_6_4
    lp = make_logprob(kind, nin, nout, seed, dense, dolog10, ypos, kw, 1.0, nograd=False)[0]

    def lnp_and_grad(x):
        xt = torch.tensor(np.asarray(x, np.float32), requires_grad=True)
        v = lp(xt, inputnumpy=False)
        g, = torch.autograd.grad(v, xt)
        return float(v), g.numpy().astype(np.float64)
    rs = np.random.RandomState(91)
    nw, nsteps, eps = 24, 4, 0.002
    var = np.linspace(0.6, 1.8, nin)                           # diagonal mass (the move's `cov`)
    coords = 0.3 * rs.standard_normal((nw, nin))
    integ = rsamp._hmc_wrapper(rs, lambda q: lnp_and_grad(q)[1], var, eps, nsteps)
    momenta = integ.cov.sample(np.random.RandomState(92), nw, nin)
    q, fac = np.zeros((nw, nin)), np.zeros(nw)
    for k in range(nw):
        q[k], fac[k] = integ((coords[k], momenta[k]))
    out["hmc_move"] = dict(case=np.array(name), coords=coords, momenta=momenta, var=var, epsilon=np.float64(eps),
                           nsteps=np.int64(nsteps), q=q, factor=fac,
                           lnp_old=np.array([lnp_and_grad(c)[0] for c in coords]),
                           lnp_new=np.array([lnp_and_grad(c)[0] for c in q]))
    print("hmc move: max |q - coords|", float(np.abs(q - coords).max()), "factor range", float(fac.min()), float(fac.max()), flush=True)


# ------------------------------------------------------------------ initial weights (torch RNG parity)
INIT_CASES = [("ChtoModelv2", 33, 33, 11), ("ChtoModelv2", 26, 457, 12), ("ChtoModelsimple", 6, 4, 13),
              ("ChtoModelv2_linear", 5, 3, 14)]


def gen_init_parity(out):
    """``torch.manual_seed(s); Model(nin, nout, None)`` and a later ``init_weight()`` of the live reference: digests
    of every tensor plus the next three draws of the global generator (the stream position afterwards)."""
    rec = {}
    for kind, nin, nout, seed in INIT_CASES:
        cls = getattr(rnn, kind)
        torch.manual_seed(seed)
        m = cls(nin, nout, None)
        tag = "%s_%d_%d" % (kind, nin, nout)
        rec[tag + "/after_construct"] = torch.rand(3).numpy()
        for k, v in m.state_dict().items():
            rec[tag + "/construct/" + k] = synth.tensor_digest(v.numpy())
        torch.manual_seed(seed + 100)
        m.init_weight()
        rec[tag + "/after_reinit"] = torch.rand(3).numpy()
        for k, v in m.state_dict().items():
            rec[tag + "/reinit/" + k] = synth.tensor_digest(v.numpy())
        print("init parity", tag, float(m.state_dict()["layer2.skip_layer.weight"].abs().max()), flush=True)
    out["init_parity"] = rec


# ------------------------------------------------------------------ 33-D README problem: a long train_NN run
def gen_train33(out, nep=None):
    """BASELINE configs[0] / README.rst:60-88: iteration 0 of ``ml_sampler`` on the 33-D Gaussian (theory =
    identity, flat priors [-5, 5], 10000 + 500 Latin-hypercube points, batch 500, ChtoModelv2(33, 33) constructed
    by the reference itself under ``torch.manual_seed``), trained by the LIVE reference's ``train_NN`` for ``nep``
    epochs with a fixed ``lr.npy``.  Stored: per-step training losses, per-epoch validation metrics, the messages of
    the epoch controller, and the residual of the trained emulator at points of the tempered posterior."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import readme33
    nep = int(os.environ.get("GOLDEN_TRAIN33_EPOCHS", "300")) if nep is None else nep
    prob = readme33.problem()
    ndim, means, cov = prob["ndim"], prob["means"], prob["cov"]
    sigma = np.sqrt(np.diag(cov))
    tmp = tempfile.mkdtemp(prefix="linna_golden33_") + "/"
    train_x, val_x = readme33.design(10000, ndim), readme33.design(500, ndim)      # frozen input design (see readme33.design)
    np.savetxt(tmp + "train_samples_x.txt", train_x); np.save(tmp + "train_samples_y.npy", train_x.copy())
    np.savetxt(tmp + "val_samples_x.txt", val_x); np.save(tmp + "val_samples_y.npy", val_x.copy())
    np.save(tmp + "lr.npy", readme33.LR)

    class _S(object):
        pass
    captured, messages = {}, []
    orig_train, orig_es = rpred.Predictor.train, rpred.EarlyStopping.step

    def spy(self, *a, **k):
        r = orig_train(self, *a, **k)
        captured["ret"], captured["pred"] = r, self
        return r

    def es_spy(self, *a, **k):
        c = orig_es(self, *a, **k)
        captured.setdefault("codes", []).append(int(c))
        return c

    import builtins
    orig_print = builtins.print

    def print_spy(*a, **k):
        msg = " ".join(str(x) for x in a).strip()
        if msg and not msg.startswith("("):
            messages.append("%d|%s" % (len(captured.get("codes", [])), msg))
        orig_print(*a, **k)
    rpred.Predictor.train, rpred.EarlyStopping.step = spy, es_spy
    rpred.print = print_spy
    import time
    t0 = time.time()
    try:
        torch.manual_seed(readme33.SEED)
        rutil.train_NN(_S(), cov, np.linalg.inv(cov), sigma, tmp, [tmp], means, None, False, True, 2, 16.0,
                       False, None, 1, rnn.ChtoModelv2, {"num_epochs": nep, "batch_size": 500}, False)
    finally:
        rpred.Predictor.train, rpred.EarlyStopping.step = orig_train, orig_es
        del rpred.print
    print("reference train_NN: %d epochs in %.0f s" % (nep, time.time() - t0), flush=True)
    train_losses, val_metrics = captured["ret"]
    pred = captured["pred"]
    # residual of the trained emulators (last and best epoch) at the posterior tempered by T = 16 (iteration 0) and T = 1
    rs = np.random.RandomState(5)
    unit = rs.standard_normal((4000, ndim))
    yinv = rutil.Y_invtransform_data(sigma, "cpu")
    rec = dict(train_digest=synth.tensor_digest(train_x), val_digest=synth.tensor_digest(val_x),
               lr=np.float64(readme33.LR), seed=np.int64(readme33.SEED), num_epochs=np.int64(nep),
               train_losses=np.asarray(train_losses, np.float64), val_metrics=np.asarray(val_metrics, np.float64),
               codes=np.asarray(captured.get("codes", []), np.int64), messages=np.array(messages))   # unit: RandomState(5)

    def residual(tag):
        for T in (16.0, 1.0):
            th = means[None, :] + np.sqrt(T) * sigma[None, :] * unit
            with torch.no_grad():
                m = yinv(pred.predict(t32(th))).numpy()
            res = (m - th) / sigma[None, :]
            rec["%s_res_rms_T%d" % (tag, T)] = np.float64(np.sqrt(np.mean(res ** 2)))
            rec["%s_res_mean_T%d" % (tag, T)] = res.mean(0)
            print("emulator (%s) residual at the T=%d posterior: rms %.3f sigma, max |mean| %.3f sigma"
                  % (tag, T, rec["%s_res_rms_T%d" % (tag, T)], np.abs(res.mean(0)).max()), flush=True)
    residual("last")
    from linna_amd import nnutils as pnnutils
    best = pnnutils.read_checkpoint(tmp + "best.pth.tar")   # weights_only=True (+ numpy scalar reconstructors for lr)
    rec["best_epoch"] = np.int64(best["epoch"])
    pred.model.load_state_dict(best["state_dict"])
    residual("best")
    out["train33_run"] = rec
    shutil.rmtree(tmp, ignore_errors=True)
    vm = np.asarray(val_metrics)
    print("train33: val loss at epochs 1/10/50/100/last:", [float(vm[min(i, len(vm) - 1), 0]) for i in (0, 9, 49, 99, len(vm) - 1)],
          "messages", messages[:12], flush=True)


# ------------------------------------------------------------------ host-side point designs
def host_design_inputs():
    """Deterministic inputs of ``gen_host_designs`` (the tests rebuild them): a synthetic 4-parameter chain that
    overshoots its prior box, and three forms of the reference's ``omegab2cut`` argument."""
    rs = np.random.RandomState(77)
    chain = rs.standard_normal((6000, 4)) * np.array([0.6, 0.5, 1.4, 0.8]) + np.array([0.3, 0.6, 0.0, 0.1])
    prior = [[-0.8, 1.4], [-0.4, 1.6], [-2.0, 2.0], [-1.0, 1.2]]
    # (a 7-element cut -- one extra parameter range -- raises IndexError in the reference: its second range is tested with
    # ``len > 6`` but read from elements 7..9, util.py:890-891; the forms that work have 4 or 10 elements)
    cuts = {"none": None, "ombh2": [0, 1, 0.01, 0.9], "ombh2_p2_p3": [0, 1, 0.01, 0.9, 2, -1.5, 1.5, 3, -0.5, 1.0]}
    return chain, prior, cuts


def importance_inputs():
    """Deterministic inputs of ``gen_importance`` (the tests rebuild them)."""
    rs = np.random.RandomState(88)
    nout, ndim, n = 7, 4, 60
    A = rs.standard_normal((nout, nout))
    cov = A @ A.T / nout + 0.3 * np.eye(nout)
    data = rs.uniform(0.5, 1.5, nout)
    theory = np.concatenate([data, [9.0, 9.0]])[None, :] + rs.standard_normal((n, nout + 2)) * 0.4    # two extra columns: the reference cuts to len(data)
    samples = rs.uniform(-1.5, 1.5, (n, ndim))
    priors = [{"param": "a", "dist": "flat", "arg1": -1.0, "arg2": 1.2}, {"param": "b", "dist": "gauss", "arg1": 0.2, "arg2": 0.7},
              {"param": "c", "dist": "flat", "arg1": -1.4, "arg2": 1.4}, {"param": "d", "dist": "gauss", "arg1": -0.3, "arg2": 1.1}]
    return data, cov, theory, samples, priors


def gen_importance(out):
    """Host helpers of the importance-sampling post step and of training-set clean-up, LIVE reference: ``LogPrior``
    (util.py:1129-1157), ``logp_theory_data`` (:1506-1517), ``chisqcut_all`` (:1260-1270: note y^T invcov y of the theory
    vector itself), ``median_absolute_deviation`` (:1308-1313)."""
    data, cov, theory, samples, priors = importance_inputs()
    invcov = np.linalg.inv(cov)
    lpr = rutil.LogPrior(priors)
    rec = dict(logprior=np.array([lpr(s) for s in samples], np.float64),
               logp=np.array(rutil.logp_theory_data(samples, theory, data, invcov, lpr), np.float64))
    tmp = tempfile.mkdtemp(prefix="linna_golden_imp_")
    y = theory[:, :len(data)]
    cutv = float(np.median([v.dot(invcov).dot(v) for v in y]))
    np.save(os.path.join(tmp, "y.npy"), y); np.savetxt(os.path.join(tmp, "x.txt"), samples)
    rutil.chisqcut_all(data, invcov, cutv, os.path.join(tmp, "y.npy"), os.path.join(tmp, "x.txt"))
    rec.update(chisqcut=np.float64(cutv), cut_y=np.load(os.path.join(tmp, "y.npy")), cut_x=np.loadtxt(os.path.join(tmp, "x.txt")))
    shutil.rmtree(tmp, ignore_errors=True)
    t = torch.tensor(theory[:, :len(data)], dtype=torch.float32)
    med = t.median(axis=0).values
    rec["mad"] = rutil.median_absolute_deviation(t, med, 0).numpy()
    out["importance_helpers"] = rec
    print("importance helpers: kept %d of %d rows; %d points outside the flat priors" % (len(rec["cut_x"]), len(samples), int(np.isinf(rec["logprior"]).sum())), flush=True)


MEANSTD_THRESHOLDS = [(0.1, 0.1), (0.02, 0.1), (0.1, 0.01), (1.0, 1.0)]


def meanstd_inputs():
    """Synthetic chains [nstep, nwalker, nparam]: stationary, drifting mean, growing spread, odd length."""
    rs = np.random.RandomState(55)
    a = rs.standard_normal((400, 6, 5))
    b = a + np.linspace(0, 0.6, 400)[:, None, None]
    c = a * np.linspace(0.7, 1.4, 400)[:, None, None]
    d = rs.standard_normal((301, 4, 3)) * np.array([1.0, 2.0, 0.5])
    return [(a, "stationary"), (b, "drift"), (c, "spread"), (d, "odd")]


def gen_host_designs(out):
    """``NN_samplerv1.gensample_chain_randomsample`` of the LIVE reference (util.py:864-897: the training points of
    iterations >= 1 are drawn from the previous chain): inside-the-prior and omega_b h^2 cuts, seed 123456."""
    chain, prior, cuts = host_design_inputs()
    ns = rutil.NN_samplerv1("/nonexistent/", prior)
    rec = {}
    for tag, cut in cuts.items():
        for n in (300, 17):
            rec["%s/%d" % (tag, n)] = np.asarray(ns.gensample_chain_randomsample(n, chain, None, omegab2cut=cut), np.float64)
    # checkmeanstd (sampler.py:370-387) of the live reference on synthetic chains: the two drift statistics it prints and
    # its verdict for thresholds on either side of them
    import contextlib
    import io
    import linna.sampler as rsamp
    for i, (chain3, _) in enumerate(meanstd_inputs()):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            verdicts = [bool(rsamp.checkmeanstd(chain3, a, b)) for a, b in MEANSTD_THRESHOLDS]
        vals = [float(x) for x in buf.getvalue().split()[:2]]
        rec["checkmeanstd/%d" % i] = np.array(vals + [float(v) for v in verdicts])
    out["host_designs"] = rec
    print("host designs", {k: v.shape for k, v in rec.items()}, flush=True)


# ------------------------------------------------------------------ ill-conditioned dense covariance
COND = dict(name="cond_26_457", kind="ChtoModelv2", nin=26, nout=457, seed=110, conds=(1e2, 1e4, 1e6), nanchor=4,
            deltas=(1e-4, 1e-3, 1e-2, 1e-1), ndir=3, temperature=1.0)


def cond_points(nin, seed, nanchor, deltas, ndir):
    """Walker positions [nanchor][1 + len(deltas) * ndir][nin]: an anchor z0 and points at growing distances from it."""
    rs = np.random.RandomState(seed + 17)
    z0 = 0.5 * rs.standard_normal((nanchor, nin))
    pts = [z0[:, None, :]]
    for dl in deltas:
        u = rs.standard_normal((nanchor, ndir, nin))
        u /= np.linalg.norm(u, axis=-1, keepdims=True)
        pts.append(z0[:, None, :] + dl * np.sqrt(nin) * u)
    return np.concatenate(pts, axis=1).astype(np.float32)


def gen_cond(out):
    """``Log_prob`` of the LIVE reference (util.py:953-955, 990-1021: fp32 ``(m - data) @ invcov @ (m - data).T``) where the
    order of summation matters: dense covariances of condition 1e2 / 1e4 / 1e6 at (26, 457), with the data vector placed so
    that the residual at each anchor is a draw from the covariance itself (chi^2 ~ nout: what a converged chain sees; the
    stiff directions then cancel against each other inside d S d^T) and points at distances 1e-4 ... 1e-1 around it (the
    stiff directions take over, chi^2 up to 1e7).  Stored: the reference's fp32 lnP and m, lnP in float64 from the
    reference's own m and the SAME fp32-rounded inverse covariance (isolates the summation error), lnP with the unrounded
    inverse, and the autograd gradient."""
    c = COND
    nin, nout, seed = c["nin"], c["nout"], c["seed"]
    z = cond_points(nin, seed, c["nanchor"], c["deltas"], c["ndir"])
    K, P = z.shape[:2]
    X_mean, X_std, y_mean, y_std = synth.transform_constants(nin, nout, seed)
    _, _, priors = synth.gaussian_problem(nin, nout, seed, dense=False)
    model = build_model(c["kind"], nin, nout, seed)
    Xt = rutil.X_transform_class(t32(X_mean), t32(X_std), "cpu", None)
    Yt = rutil.Y_transform_class(t32(y_mean), t32(y_std), "cpu", ypositive=False)
    pred = rpred.Predictor(nin, nout, model=model, X_transform=Xt, y_transform=Yt, device="cpu")
    transform = rutil.Transform(priors)
    rs = np.random.RandomState(seed + 23)
    xi = rs.standard_normal((K, nout))
    rec = dict(z=z, conds=np.array(c["conds"]), xi=xi)
    for ci, cond in enumerate(c["conds"]):
        cov, inv, half = synth.cond_problem(nin, nout, seed, cond)
        sigma = np.sqrt(np.diag(cov))
        yinv = rutil.Y_invtransform_data(sigma, "cpu")
        inv32 = inv.astype(np.float32)
        data = np.zeros((K, nout)); m = np.zeros((K, P, nout), np.float32)
        l32 = np.zeros((K, P), np.float32); l64 = np.zeros((K, P)); l64u = np.zeros((K, P)); gr = np.zeros((K, P, nin), np.float32)
        for k in range(K):
            theta = np.stack([np.atleast_1d(transform(zi)) for zi in z[k]]).astype(np.float32)
            with torch.no_grad():
                mk = np.stack([yinv(pred.predict(t32(th))[None, :])[0].numpy() for th in theta])
            data[k] = mk[0].astype(np.float64) - half @ xi[k]                 # residual at the anchor: a draw from the covariance
            lp = rutil.Log_prob(t32(data[k]), torch.from_numpy(inv32), pred, yinv, transform, c["temperature"],
                                rutil.gaussianlogliklihood, nograd=True)
            lpg = rutil.Log_prob(t32(data[k]), torch.from_numpy(inv32), pred, yinv, transform, c["temperature"],
                                 rutil.gaussianlogliklihood, nograd=False)
            d32 = data[k].astype(np.float32)
            for i, zi in enumerate(z[k]):
                l32[k, i] = float(lp(zi))
                x = t32(zi).clone().requires_grad_()
                gr[k, i] = torch.autograd.grad(lpg(x, inputnumpy=False), x)[0].numpy()
                dd = mk[i].astype(np.float64) - d32.astype(np.float64)
                prior = -0.5 * float(np.sum(zi.astype(np.float64) ** 2))
                l64[k, i] = -0.5 * dd @ inv32.astype(np.float64) @ dd / c["temperature"] + prior
                l64u[k, i] = -0.5 * dd @ inv @ dd / c["temperature"] + prior
            m[k] = mk
        for key, v in (("data", data), ("m", m), ("lnP32", l32), ("lnP64", l64), ("lnP64_exact_inv", l64u), ("grad", gr)):
            rec["%s/%d" % (key, ci)] = v
        e = np.abs(l32 - l64)
        print("cond %.0e: chi2 at anchors %s; |lnP32 - lnP64| anchors max %.3g, all points max rel %.3g" % (
            cond, np.round(-2 * (l64[:, 0] + 0.5 * np.sum(z[:, 0].astype(np.float64) ** 2, -1))), e[:, 0].max(), (e / np.abs(l64)).max()), flush=True)
    out[c["name"]] = rec


GENERATORS = [("cond", gen_cond), ("host_designs", gen_host_designs), ("importance", gen_importance), ("fixture", gen_fixture), ("serving", gen_serving), ("training", gen_training), ("train_nn", gen_train_nn), ("train_nn_ypos", gen_train_nn_ypos), ("train_nn_usebest", gen_train_nn_usebest),
              ("loader_order", gen_loader_order), ("early_stopping", gen_early_stopping), ("hmc", gen_hmc), ("hmc_move", gen_hmc_move), ("callbacks", gen_callbacks),
              ("init_parity", gen_init_parity), ("train33", gen_train33)]


def main():
    """``make_golden.py`` rewrites everything; ``make_golden.py init_parity train33`` only the named groups."""
    want = sys.argv[1:]
    out = {}
    for name, fn in GENERATORS:
        if not want or name in want:
            fn(out)
    for name, rec in out.items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
        sz = os.path.getsize(os.path.join(HERE, name + ".npz"))
        print("wrote %s.npz (%d KB)" % (name, sz // 1024))


if __name__ == "__main__":
    main()
