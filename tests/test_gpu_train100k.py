"""GPU: BASELINE configs[2] at its stated size -- a RESIDENT training set of 100 000 rows (``yamlfile/training_3x2pt.yaml:34-39``;
the arrays ``train_NN`` concatenates over iterations, util.py:1346-1373), ``ChtoModelv2(26, 457)``, dense covariance, batch 500:
``linna_loss_targets`` over the whole set once, then ``linna_net_train_step_update`` (gather by row index -> forward -> loss ->
backward -> AdamW, one C call) over a full shuffled epoch of 200 steps.  Row indices run past 2^16.

Checks: (1) with the optimiser frozen (lr = 0, weight decay = 0) every step's loss is the oracle's loss of exactly those
rows (a sample of steps, the rows with the largest indices included); (2) a real epoch from the same start trains
(finite, decreasing), and its first steps are bit-identical to the same batches gathered from a 500-row copy of those
rows -- where a row lives in the resident set changes nothing.
"""
import numpy as np
import pytest
import torch

from linna_amd import _lib

pytestmark = pytest.mark.gpu

NROWS, NIN, NOUT, B = 100000, 26, 457, 500


def _problem():
    rs = np.random.RandomState(5)
    q, _ = np.linalg.qr(rs.standard_normal((NOUT, NOUT)))
    cov = (q * (np.logspace(0, -2, NOUT) * 0.1)[None, :]) @ q.T
    cov = 0.5 * (cov + cov.T)
    data, sigma = rs.uniform(size=NOUT), np.sqrt(np.diag(cov))
    X_mean, X_std = rs.uniform(-0.5, 0.5, NIN).astype(np.float32), rs.uniform(0.5, 3.0, NIN).astype(np.float32)
    y_mean, y_std = rs.uniform(-0.5, 0.5, NOUT).astype(np.float32), rs.uniform(0.5, 2.0, NOUT).astype(np.float32)
    A = (0.3 * rs.standard_normal((NOUT, NIN))).astype(np.float32)
    X = (X_mean[None, :] + X_std[None, :] * rs.standard_normal((NROWS, NIN))).astype(np.float32)
    # a smooth map of the parameters plus noise, in units of sigma around the data vector (what a theory code returns)
    Y = (data[None, :] + sigma[None, :] * (np.tanh((X - X_mean) / X_std) @ A.T + 0.3 * rs.standard_normal((NROWS, NOUT)))).astype(np.float32)
    Y[77777, 5] = 1e10                                        # sentinels the loss masks (util.py:1072), on a row past 2^16
    Y[99999, 0] = 1e-30
    return dict(cov=cov, data=data, sigma=sigma, X_mean=X_mean, X_std=X_std, y_mean=y_mean, y_std=y_std, X=X, Y=Y)


def _engine(p, X, Y, seed=1234):
    from linna_amd import nn, util, predictor_gpu, trainer
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32))
    torch.manual_seed(seed)
    model = nn.ChtoModelv2(NIN, NOUT, None)
    pred = predictor_gpu.Predictor(NIN, NOUT, model=model, device="cuda",
                                   X_transform=util.X_transform_class(t(p["X_mean"]), t(p["X_std"]), "cpu", None),
                                   y_transform=util.Y_transform_class(t(p["y_mean"]), t(p["y_std"]), "cpu"))
    ytd = util.Y_transform_data(p["sigma"], "cpu")
    yinv = util.Y_invtransform_class(t(p["y_mean"]), t(p["y_std"]), t(p["data"]), "cpu")
    lf = util.Loss_fn(t(p["data"]), torch.tensor(p["cov"], dtype=torch.float64), torch.tensor(np.linalg.inv(p["cov"]), dtype=torch.float64),
                      ytd, yinv, "cpu")
    loader = predictor_gpu.BatchLoader(util.ArrayDataset(X, Y), B, shuffle=True, drop_last=True)
    eng = trainer.TrainEngine(pred, loader, lf, None)
    return model, eng, loader, lf


def test_full_epoch_over_a_100k_row_resident_set():
    from oracle import emulator, training
    from linna_amd.predictor_gpu import _AdamWState
    p = _problem()
    model, eng, loader, lf = _engine(p, p["X"], p["Y"])
    assert eng.X.shape == (NROWS, NIN) and eng.Y.shape == (NROWS, NOUT)
    torch.manual_seed(99)
    order = loader.epoch_rows()                                # [200, 500] int32: one shuffled epoch (DataLoader order)
    assert order.shape == (NROWS // B, B) and len(np.unique(order)) == NROWS and order.max() == NROWS - 1
    perm = torch.from_numpy(np.ascontiguousarray(order)).to("cuda")
    nsteps = len(order)
    w0 = {k: v.cpu().numpy().copy() for k, v in model.state_dict().items()}
    sigma, ymean, ystd, data_norm, cinv = lf.auxileryfunction.arrays()
    stats = dict(X_mean=p["X_mean"], X_std=p["X_std"], y_mean=ymean, y_std=ystd, sigma=sigma, data_norm=data_norm, icov_norm=cinv)

    # (1) frozen optimiser: 200 steps through linna_net_train_step_update, parameters unchanged, losses = oracle's
    frozen = _AdamWState(model, 0.0, weight_decay=0.0)
    hist = torch.zeros(nsteps, dtype=torch.float32, device="cuda")
    for s in range(nsteps):
        eng.step(frozen, perm[s], hist[s:s + 1])
    torch.cuda.synchronize()
    assert eng.one_update is True                              # the one-call step (two launches: forward + loss + dX chain, parameter gradients + AdamW) is what ran
    assert eng.YN.shape[0] == NROWS                            # linna_loss_targets over the whole resident set
    for k, v in model.state_dict().items():
        np.testing.assert_array_equal(v.cpu().numpy(), w0[k])
    got = hist.cpu().numpy()
    assert np.all(np.isfinite(got))
    big = int(np.argmax(order.max(axis=1) == NROWS - 1))        # the step that holds row 99 999 (and its 1e-30 sentinel)
    s77 = int(np.where((order == 77777).any(axis=1))[0][0])     # ... and the one with the 1e10 sentinel
    for s in sorted({0, 1, nsteps // 2, nsteps - 1, big, s77}):
        rows = order[s]
        x = (p["X"][rows] - p["X_mean"][None, :]) / p["X_std"][None, :]
        predo = emulator.forward(w0, x.astype(np.float32), "ChtoModelv2", NIN, NOUT)
        ref = training.loss(predo, p["Y"][rows], data_norm, cinv, sigma, ymean, ystd)
        np.testing.assert_allclose(got[s], ref, rtol=3e-6, err_msg="step %d (max row %d)" % (s, rows.max()))
    # the normalised targets of the masked elements are NaN, everything else finite (linna_loss_targets)
    yn = eng.YN[:, :NOUT]
    assert bool(torch.isnan(yn[77777, 5])) and bool(torch.isnan(yn[99999, 0])) and int(torch.isnan(yn).sum()) == 2

    # (2) a real epoch trains; where a row lives changes nothing: first steps == the same batches from a 500-row copy
    opt = _AdamWState(model, 1e-3, weight_decay=1e-4)
    for s in range(nsteps):
        eng.step(opt, perm[s], hist[s:s + 1])
    torch.cuda.synchronize()
    tr = hist.cpu().numpy()
    assert np.all(np.isfinite(tr)) and tr[-20:].mean() < 0.5 * tr[:5].mean(), (tr[:5], tr[-20:])
    after3 = None
    model2, eng2, _, _ = _engine(p, p["X"][order[:3].reshape(-1)], p["Y"][order[:3].reshape(-1)])
    for k, v in model2.state_dict().items():
        np.testing.assert_array_equal(v.cpu().numpy(), w0[k])   # same seed, same initial weights
    opt2 = _AdamWState(model2, 1e-3, weight_decay=1e-4)
    h2 = torch.zeros(3, dtype=torch.float32, device="cuda")
    for s in range(3):
        eng2.step(opt2, torch.arange(s * B, (s + 1) * B, dtype=torch.int32, device="cuda"), h2[s:s + 1])
    torch.cuda.synchronize()
    np.testing.assert_array_equal(h2.cpu().numpy(), tr[:3])    # bit for bit: gather by index 0..499 == gather by index up to 99 999
