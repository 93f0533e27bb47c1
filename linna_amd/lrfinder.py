"""Learning-rate range test on the HIP train step.

The reference calls the third-party ``torch_lr_finder.LRFinder.range_test`` (not in its
tree; predictor_gpu.py:222-238) with ``start 1e-4 -> end 5e-3, 100 iterations, val_loader``
and then picks the learning rate of steepest loss descent.  This restates that package's
published algorithm (exponential schedule, validation loss per iteration, exponential
smoothing 0.05, divergence threshold 5) on the native training step -- PARITY UNPINNED for
the third-party part; the selection rule (:232-238) is the reference's.
"""
import numpy as np
import torch

from .predictor_gpu import _AdamWState


def range_test(pred, engine, start_lr=1e-4, end_lr=5e-3, num_iter=100, smooth_f=0.05, diverge_th=5.0, history=None):
    """``history``: a dict that receives the recorded curve (``lr``, ``loss``), as ``LRFinder.history`` holds it."""
    model = pred.model
    saved = model.flat_params().clone()
    rng = torch.get_rng_state()
    opt = _AdamWState(model, start_lr, weight_decay=1e-4)
    lrs, losses, best = [], [], None
    batches, pos = [], 0
    for it in range(num_iter):
        if pos >= len(batches):
            batches, pos = engine.loader.epoch_batches(), 0
        rows = batches[pos].to(torch.int32).to(engine.dev)
        pos += 1
        opt.lr = start_lr * (end_lr / start_lr) ** (it / max(num_iter - 1, 1))
        opt.push_hyper()
        engine.rows.copy_(rows)
        engine._step_body(opt, local=True)
        if engine.val is not None:
            engine.validate()
            loss = float(engine.val["loss_rows"].mean().item())
        else:
            loss = float(engine.loss_mean.item())
        lrs.append(opt.lr)
        if it == 0:
            best = loss
        else:
            if smooth_f > 0:
                loss = smooth_f * loss + (1 - smooth_f) * losses[-1]
            best = min(best, loss)
        losses.append(loss)
        if not np.isfinite(loss) or loss > diverge_th * best:
            break
    model.flat_params().copy_(saved)
    torch.set_rng_state(rng)
    if history is not None:
        history["lr"], history["loss"] = list(lrs), list(losses)
    if len(losses) < 2:
        return start_lr
    lr = lrs[int(np.gradient(np.array(losses)).argmin())]          # predictor_gpu.py:234-235
    if lr > 1e0:
        lr = lr / 1e2
    return float(lr)
