"""Oracle: training statistics, chi^2-ratio loss, analytic gradients, AdamW (numpy).

TEST INFRASTRUCTURE ONLY.  Restates linna/util.py:1055-1127 (Auxilleryfunc, Loss_fn,
Val_metric_fn), :1308-1313, :1410-1460 (train_NN statistics) and the torch.optim.AdamW
update used at linna/predictor_gpu.py:267,287.  ``lr_range_test`` (last function) restates the
third-party learning-rate finder the reference calls at predictor_gpu.py:222-238 (torch_lr_finder:
absent from the reference tree and from this image) from its published algorithm -- PARITY UNPINNED
for that function; the selection rule behind it is the reference's.
"""
import numpy as np

from . import emulator


def lower_median(a, axis=0):
    """torch.median semantics (util.py:1313,1449; predictor_gpu.py:62): for an even count
    the LOWER of the two middle values, not their mean."""
    a = np.asarray(a)
    s = np.sort(a, axis=axis)
    n = a.shape[axis]
    return np.take(s, (n - 1) // 2, axis=axis)


def ypositive_clip(train_x, train_y, train_y_last, val_x, val_y):
    """util.py:1410-1431 (``ypositive=True``): clip to [1e-30, 1e10] and delete the rows that are 1e-30 throughout -- with
    the reference's loop, which uses the indices found before the first deletion.  Returns copies."""
    train_x, val_x = np.array(train_x), np.array(val_x)
    train_y, train_y_last, val_y = (np.array(a, np.float64) for a in (train_y, train_y_last, val_y))
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
    return train_x, train_y, train_y_last, val_x, val_y


def data_statistics(train_x, train_y, train_y_last, sigma, dolog10index=None, ypositive=False):
    """X_mean/X_std, y_mean/y_std of util.py:1433-1451.

    ``ypositive=False``: sentinel clipping (util.py:1433-1438) is applied to copies; y statistics of the first iteration's
    targets.  ``ypositive=True`` (arrays already through ``ypositive_clip``): lower median / median absolute deviation of
    log(y / sigma) over ALL training rows, no floor on y_std (util.py:1444-1447).  Returns float32 arrays.
    """
    if ypositive:
        train_y = np.array(train_y, np.float64)
    else:
        train_y = np.clip(np.array(train_y, np.float64), -1e5, 1e10)
        train_y_last = np.clip(np.array(train_y_last, np.float64), -1e5, 1e10)
    X1 = np.array(train_x, np.float32)
    if dolog10index is not None:
        for i in dolog10index:
            X1[:, i] = np.log10(X1[:, i])
    X_mean = X1.mean(axis=0, dtype=np.float32)
    X_std = X1.std(axis=0, ddof=1, dtype=np.float32)              # torch .std is unbiased
    if ypositive:
        ys = np.log(train_y.astype(np.float32) / np.asarray(sigma, np.float32)[None, :])
        y_mean = lower_median(ys, 0)
        y_std = lower_median(np.abs(ys - y_mean[None, :]), 0)
        return X_mean, X_std.astype(np.float32), y_mean.astype(np.float32), y_std.astype(np.float32)
    ys = train_y_last.astype(np.float32) / np.asarray(sigma, np.float32)[None, :]
    y_mean = lower_median(ys, 0)
    y_std = lower_median(np.abs(ys - y_mean[None, :]), 0)
    y_std = np.where(y_std < 1e-10, np.float32(1.0), y_std)       # util.py:1451
    return X_mean, X_std.astype(np.float32), y_mean.astype(np.float32), y_std.astype(np.float32)


def normalised_inverse_cov(cov, sigma, y_std, ypositive=False, data=None):
    """util.py:1063-1064 with util.py:447 and :590: C~ = D2 (D1 cov^T D1)^T D2 in fp64 with
    D1 = diag(1/sigma_f32), D2 = diag(1/y_std_f32); returns inverse cast to fp32.
    ``ypositive`` (util.py:579-585): between the two scalings the matrix becomes log(1 + E c E), E = diag(1 / data) with
    the fp32 data vector in PHYSICAL units as the reference passes it, entries <= -1 set to 1e-10 - 1 first."""
    d1 = 1.0 / np.asarray(sigma, np.float32).astype(np.float64)
    d2 = 1.0 / np.asarray(y_std, np.float32).astype(np.float64)
    c1 = (np.diag(d1) @ np.asarray(cov, np.float64).T) @ np.diag(d1).T
    if ypositive:
        e = np.diag(1.0 / np.asarray(data, np.float32).astype(np.float64))
        c0 = (e @ c1.T) @ e.T
        c0[c0 <= -1] = 1e-10 - 1
        c1 = np.log(1 + c0)
    c2 = (np.diag(d2) @ c1.T) @ np.diag(d2).T
    return np.linalg.inv(c2).astype(np.float32)


def normalise_target(y, sigma, y_mean, y_std, ypositive=False):
    """y_inv_transform(y_transform_data(y)) (util.py:1071): ((y/sigma) - y_mean)/y_std; ``ypositive`` (util.py:567-571):
    log(y/sigma) in place of y/sigma."""
    y = np.asarray(y, np.float32)
    v = y / np.asarray(sigma, np.float32)[None, :]
    if ypositive:
        with np.errstate(invalid="ignore", divide="ignore"):
            v = np.log(v)
    return (v - y_mean[None, :]) / y_std[None, :]


def normalise_data(data, sigma, y_mean, y_std, ypositive=False):
    """util.py:1069: the data vector in the network's output space, NaN -> 1e-30 (the mask sentinel of util.py:1072)."""
    d = normalise_target(np.asarray(data, np.float32)[None, :], sigma, y_mean, y_std, ypositive)[0]
    return np.where(np.isnan(d), np.float32(1e-30), d).astype(np.float32)


def aux(pred, target, data_norm, icov_norm, sigma, y_mean, y_std, ypositive=False):
    """util.py:1070-1088.  ``pred`` is the raw network output (normalised space),
    ``target`` is the physical data vector batch, ``data_norm`` the normalised data.

    Returns loss[B], chisqMd[B], chisqnnd[B], plus (delta, notmask) used by ``loss_grad``.
    """
    dt = pred.dtype
    target = np.asarray(target, dt)
    tnorm = normalise_target(target, sigma, y_mean, y_std, ypositive).astype(dt)
    mask = (target == dt.type(1e-30)) | (target == dt.type(1e10)) | (data_norm[None, :] == dt.type(1e-30))
    C = icov_norm.astype(dt)

    def chisq(delta):
        delta = np.where(mask, dt.type(0), delta)
        return ((delta @ C) * delta).sum(-1), delta

    chisqnnd, _ = chisq(pred - data_norm[None, :])
    chisqMd, _ = chisq(tnorm - data_norm[None, :])
    chisqMnn, delta = chisq(tnorm - pred)
    floor = dt.type(0.5 * target.shape[1])
    chisqMd = np.where(chisqMd < floor, floor, chisqMd)             # util.py:1086
    return chisqMnn / chisqMd, chisqMd, chisqnnd, delta, ~mask


def loss(pred, target, data_norm, icov_norm, sigma, y_mean, y_std, ypositive=False):
    """util.py:1105-1116: mean over the batch."""
    return aux(pred, target, data_norm, icov_norm, sigma, y_mean, y_std, ypositive)[0].mean(dtype=pred.dtype)


def val_metric(pred, target, data_norm, icov_norm, sigma, y_mean, y_std, ypositive=False):
    """util.py:1124-1127: [median(loss), max|chisqnnd/chisqMd - 1|, median(same)]."""
    l, cMd, cnnd, _, _ = aux(pred, target, data_norm, icov_norm, sigma, y_mean, y_std, ypositive)
    frac = np.abs(cnnd / cMd - 1)
    return np.array([lower_median(l), frac.max(), lower_median(frac)], np.float32)


def loss_grad(pred, target, data_norm, icov_norm, sigma, y_mean, y_std, ypositive=False):
    """d mean(loss) / d pred: -(delta (C + C^T)) / (B * chisqMd), zero where masked."""
    l, cMd, _, delta, notmask = aux(pred, target, data_norm, icov_norm, sigma, y_mean, y_std, ypositive)
    C = icov_norm.astype(pred.dtype)
    g = -(delta @ C + delta @ C.T) / (pred.dtype.type(pred.shape[0]) * cMd[:, None])
    return l.mean(dtype=pred.dtype), np.where(notmask, g, 0).astype(pred.dtype)


def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-4):
    """One torch.optim.AdamW update on float32 arrays (in place); ``step`` is 1-based."""
    f = np.float32
    p *= f(1 - lr * weight_decay)
    m += (g - m) * f(1 - beta1)
    v *= f(beta2)
    v += f(1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = np.sqrt(v) / f(np.sqrt(bc2)) + f(eps)
    p -= f(lr / bc1) * (m / denom)
    return p, m, v


def train_step(params, opt_state, X, y, stats, kind, in_size, out_size, lr, weight_decay=1e-4,
               **topo_kw):
    """One minibatch of predictor_gpu.py:273-288: forward -> loss -> backward -> AdamW.

    ``stats`` = dict(X_mean, X_std, y_mean, y_std, sigma, data_norm, icov_norm).
    ``opt_state`` = dict(step:int, m:{key:arr}, v:{key:arr}).  Returns (loss, grads).
    """
    x = (np.asarray(X, np.float32) - stats["X_mean"][None, :]) / stats["X_std"][None, :]
    pred, caches = emulator.forward(params, x, kind, in_size, out_size, keep=True, **topo_kw)
    l, dpred = loss_grad(pred, y, stats["data_norm"], stats["icov_norm"], stats["sigma"],
                         stats["y_mean"], stats["y_std"], stats.get("ypositive", False))
    _, grads = emulator.backward(params, caches, dpred, kind, in_size, out_size, **topo_kw)
    opt_state["step"] += 1
    for k in params:
        g = grads.get(k)
        if g is None:
            continue
        adamw_step(params[k], g.astype(np.float32), opt_state["m"][k], opt_state["v"][k],
                   opt_state["step"], lr, weight_decay=weight_decay)
    return l, grads


def new_opt_state(params):
    return {"step": 0, "m": {k: np.zeros_like(v) for k, v in params.items()},
            "v": {k: np.zeros_like(v) for k, v in params.items()}}


def lr_range_test(params, batches, val, stats, kind, in_size, out_size, start_lr=1e-4, end_lr=5e-3, num_iter=100,
                  smooth_f=0.05, diverge_th=5.0, weight_decay=1e-4, **topo_kw):
    """The learning-rate range test predictor_gpu.py:222-238 runs before training: ``torch_lr_finder.LRFinder
    .range_test(dataset, val_loader=val_dataset, end_lr=5e-3, num_iter=100)`` on a COPY of the model with
    AdamW(lr=1e-4, weight_decay=1e-4), then the learning rate of steepest descent of the recorded loss curve.
    torch_lr_finder (third party, requirements.txt; absent from the reference tree and from this image) is restated
    from its published algorithm -- PARITY UNPINNED for that part: per iteration one optimiser step on the next
    training batch at lr_i = start (end / start)^(i / (num_iter - 1)), the loss over the validation set (sample-weighted
    mean of the batch losses), exponential smoothing ``smooth_f`` against the previous RECORDED value, stop once the
    loss exceeds ``diverge_th`` x the best.  ``batches``: [(X, y)] cycled; ``val``: [(X, y)].  Returns (lr, lrs, losses)."""
    params = {k: v.copy() for k, v in params.items()}
    opt = new_opt_state(params)
    lrs, losses, best = [], [], None
    for it in range(num_iter):
        lr = start_lr * (end_lr / start_lr) ** (it / max(num_iter - 1, 1))
        X, y = batches[it % len(batches)]
        train_step(params, opt, X, y, stats, kind, in_size, out_size, lr, weight_decay=weight_decay, **topo_kw)
        tot, cnt = 0.0, 0
        for Xv, yv in val:
            x = (np.asarray(Xv, np.float32) - stats["X_mean"][None, :]) / stats["X_std"][None, :]
            pred = emulator.forward(params, x, kind, in_size, out_size, **topo_kw)
            tot += float(loss(pred, yv, stats["data_norm"], stats["icov_norm"], stats["sigma"], stats["y_mean"], stats["y_std"])) * len(Xv)
            cnt += len(Xv)
        l = tot / cnt
        lrs.append(lr)
        if it == 0:
            best = l
        else:
            if smooth_f > 0:
                l = smooth_f * l + (1 - smooth_f) * losses[-1]
            best = min(best, l)
        losses.append(l)
        if not np.isfinite(l) or l > diverge_th * best:
            break
    if len(losses) < 2:
        return start_lr, lrs, losses
    lr = lrs[int(np.gradient(np.array(losses)).argmin())]          # predictor_gpu.py:234-235
    if lr > 1e0:
        lr = lr / 1e2
    return float(lr), lrs, losses
