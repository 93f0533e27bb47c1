"""HIP training engine behind ``Predictor.train`` (predictor_gpu.py:201-449 of the reference).

Data layout: the whole training / validation set lives in HBM once (``X[n, nin]``,
``Y[n, nout]`` fp32); a minibatch is an int32 index vector.  One optimiser step is

    gather + input transform -> network forward (activations kept) -> chi^2-ratio loss and its
    gradient -> network backward (flat gradient buffer) -> [RCCL all-reduce over ranks] -> AdamW

all enqueued on one HIP stream as direct launches (``TrainEngine(use_graph=True)`` captures the same
step once into a hipGraph and replays it -- learning rate / weight decay are read from a device
array, the AdamW step counter lives on the device -- measured equal or slower on ROCm 7.2, so
``run`` does not use it).  The per-epoch controller is the reference's, on the host.
"""
import ctypes as C
import os
import time

import numpy as np
import torch

from . import _lib
from . import nnutils
from .predictor_gpu import EarlyStopping, _AdamWState, _lower_median


class TrainEngine(object):
    def __init__(self, pred, loader, loss_fn, val_loader, world_size=1, dist_group=None, use_graph=False):
        self.pred, self.model = pred, pred.model
        dev = self.model.device
        if dev.type != "cuda":
            raise _lib.LinnaHipError("training runs on the GPU only (no CPU fallback)")
        self.dev, self.ctx = dev, _lib.ctx(dev.index)
        self.world, self.group = int(world_size), dist_group
        self.use_graph = use_graph and self.world == 1
        self.nin, self.nout = self.model.in_size, self.model.out_size
        f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32), device=dev)
        self.X, self.Y = f32(loader.dataset.X), f32(loader.dataset.y)
        self.B = loader.batch_size
        self.loader = loader
        self.k = pred._device_consts()
        sigma, ymean, ystd, data_norm, cinv = loss_fn.auxileryfunction.arrays()
        ldc = _lib.ld4(self.nout)         # padded rows: streamable by 16-byte LDS-DMA
        cinv = np.pad(cinv, ((0, 0), (0, ldc - self.nout)))
        self._keep = dict(sigma=f32(sigma), ymean=f32(ymean), ystd=f32(ystd), data_norm=f32(data_norm), cinv=f32(cinv))
        d = _lib.LossDesc()
        d.nout = self.nout
        d.sigma, d.ymean, d.ystd = (_lib.ptr(self._keep[n]) for n in ("sigma", "ymean", "ystd"))
        d.data_norm, d.Cinv, d.ldc = _lib.ptr(self._keep["data_norm"]), _lib.ptr(self._keep["cinv"]), ldc
        d.ylog = 1 if getattr(loss_fn.auxileryfunction.y_inv_transform, "ypositive", False) else 0     # util.py:567-571
        self.desc = d
        self.den = self._chi2_md(self.Y)
        self.val = None
        if val_loader is not None:
            VX, VY = f32(val_loader.dataset.X), f32(val_loader.dataset.y)
            self.val = dict(X=VX, Y=VY, den=self._chi2_md(VY), n=VX.shape[0])
        B, ldx, ldo = self.B, _lib.ld4(self.nin), _lib.ld4(self.nout)
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        self.xb, self.predb, self.dpred = z(B, ldx), z(B, ldo), z(B, ldo)
        self.scratch = z(_lib.load().linna_loss_scratch_bytes(B, self.nout) // 4 + 4)
        self.loss_rows, self.loss_mean = z(B), z(1)
        if self.world > 1:
            self.loss_mean = self.model.grad_tail()      # right behind the gradients: ONE all-reduce carries both
        self.rows = torch.zeros(B, dtype=torch.int32, device=dev)
        self.graph = None
        self.one_launch = None                           # None: try linna_net_forward_loss on the first step
        self.one_update = None                           # None: try linna_net_train_step_update (one rank) on the first step
        self._updated = False
        self.YN = None
        self.inv_batch = 1.0 / (B * self.world)          # the global batch is B per rank x ranks
        _lib.call("linna_net_prepare", self.model.net_handle(with_grads=True), 1, 0)   # no allocation on the launch path
        _lib.call("linna_net_prepare_loss", self.model.net_handle(with_grads=True), C.byref(self.desc))

    def _chi2_md(self, Y):
        n = Y.shape[0]
        den = torch.empty(n, dtype=torch.float32, device=self.dev)
        scratch = torch.empty(_lib.load().linna_loss_scratch_bytes(n, self.nout) // 4 + 4, dtype=torch.float32, device=self.dev)
        _lib.call("linna_chi2_md", self.ctx, C.byref(self.desc), _lib.ptr(Y), Y.stride(0), n, _lib.ptr(scratch),
                  _lib.ptr(den), _lib.stream())
        torch.cuda.current_stream().synchronize()
        return den

    # ------------------------------------------------------------------ one optimiser step
    def _forward_loss_backward(self, rows=None, loss_out=None, opt=None, update=False):
        """``rows`` / ``loss_out``: device int32 row indices and a 1-float device slot for the mean loss; default the
        engine's own fixed buffers (what a captured graph needs).  Direct launches pass the caller's tensors and save
        two copy kernels per step."""
        k, st = self.k, _lib.stream()
        rows = self.rows if rows is None else rows
        loss_out = self.loss_mean if loss_out is None else loss_out
        if self.one_launch is not False:
            # gather + transform + forward + loss + d loss / d pred in ONE launch when the network and the loss fit the
            # whole-network kernel (seven launches otherwise)
            # (linna_net_train_step: that launch AND the backward in one call, the batch mean of the loss and AdamW's step
            # constants riding in the backward's dX-chain launch; `update`: AdamW too, in the epilogue of the grouped
            # parameter-gradient launch -- the whole optimiser step in one call; since round 3 forward + loss + dX chain are ONE
            # launch on the small-batch engines: two launches per step)
            m = self.model
            self._updated = False
            if update and opt is not None and self.one_update is not False:
                rc = _lib.load().linna_net_train_step_update(
                    m.net_handle(with_grads=True), C.byref(self.desc), _lib.ptr(self.X), self.X.stride(0), _lib.iptr(rows), self.B,
                    _lib.iptr(k["lg"]) if k["lg"] is not None else None, _lib.ptr(k["xmean"]), _lib.ptr(k["xstd"]), _lib.ptr(self.xb),
                    self.xb.stride(0), _lib.ptr(m.workspace(self.B)), _lib.ptr(self.predb), self.predb.stride(0), _lib.ptr(self._targets()),
                    self.YN.stride(0), _lib.ptr(self.den), self.inv_batch, _lib.ptr(self.loss_rows), _lib.ptr(loss_out), _lib.ptr(self.dpred),
                    self.dpred.stride(0), _lib.ptr(m.workspace(self.B, "bwd")), _lib.ptr(m._flat), _lib.ptr(opt.m), _lib.ptr(opt.v),
                    m._flat.numel(), _lib.ptr(opt.hyper), _lib.iptr(opt.step_dev), opt.betas[0], opt.betas[1], opt.eps, st)
                if rc == 0:
                    self.one_launch = self.one_update = self._updated = True
                    self._prepared = True
                    m._last_input = self.xb
                    return
                if rc != _lib.ERR_UNSUPPORTED or self.one_update is True:
                    _lib.check(rc)
                self.one_update = False
            rc = _lib.load().linna_net_train_step(
                m.net_handle(with_grads=True), C.byref(self.desc), _lib.ptr(self.X), self.X.stride(0), _lib.iptr(rows), self.B,
                _lib.iptr(k["lg"]) if k["lg"] is not None else None, _lib.ptr(k["xmean"]), _lib.ptr(k["xstd"]), _lib.ptr(self.xb),
                self.xb.stride(0), _lib.ptr(m.workspace(self.B)), _lib.ptr(self.predb), self.predb.stride(0), _lib.ptr(self._targets()),
                self.YN.stride(0), _lib.ptr(self.den), self.inv_batch, _lib.ptr(self.loss_rows), _lib.ptr(loss_out), _lib.ptr(self.dpred),
                self.dpred.stride(0), _lib.ptr(m.workspace(self.B, "bwd")), _lib.ptr(opt.hyper) if opt is not None else None,
                _lib.iptr(opt.step_dev) if opt is not None else None, opt.betas[0] if opt is not None else 0.0,
                opt.betas[1] if opt is not None else 0.0, st)
            if rc == 0:
                self.one_launch = True
                self._prepared = opt is not None
                m._last_input = self.xb
                return
            if rc != _lib.ERR_UNSUPPORTED or self.one_launch is True:
                _lib.check(rc)
            self.one_launch = False
        _lib.call("linna_gather_xform", self.ctx, _lib.ptr(self.X), self.X.stride(0), _lib.iptr(rows), self.B, self.nin,
                  _lib.iptr(k["lg"]) if k["lg"] is not None else None, _lib.ptr(k["xmean"]), _lib.ptr(k["xstd"]),
                  _lib.ptr(self.xb), self.xb.stride(0), st)
        self.model.forward_buffer(self.xb, self.B, out=self.predb)
        _lib.call("linna_chi2_ratio_loss_fwd_bwd", self.ctx, C.byref(self.desc), _lib.ptr(self.predb), self.predb.stride(0),
                  _lib.ptr(self.Y), self.Y.stride(0), _lib.ptr(self.den), _lib.iptr(rows), self.B,
                  _lib.ptr(self.scratch), _lib.ptr(self.loss_rows), _lib.ptr(loss_out), _lib.ptr(self.dpred),
                  self.dpred.stride(0), self.inv_batch, st)
        self.model.backward(self.dpred[:, :self.nout], param_grads=True)

    def _targets(self):
        """Normalised targets of the whole training set with their mask (``linna_loss_targets``), computed once."""
        if self.YN is None:
            self.YN = torch.empty((self.Y.shape[0], _lib.ld4(self.nout)), dtype=torch.float32, device=self.dev)
            _lib.call("linna_loss_targets", self.ctx, C.byref(self.desc), _lib.ptr(self.Y), self.Y.stride(0), self.Y.shape[0],
                      _lib.ptr(self.YN), self.YN.stride(0), _lib.stream())
        return self.YN

    def _step_body(self, opt, rows=None, loss_out=None, local=False):
        """``local``: this rank alone (no collective, gradient of its own batch) -- the learning-rate range test, which
        the reference runs on rank 0 only, on a private copy of the model (predictor_gpu.py:223-227)."""
        self._prepared = self._updated = False
        if self.world > 1 and not local:
            from . import dist as ldist
            self._forward_loss_backward(rows, self.loss_mean, opt)
            ldist.allreduce_grads(self.model.flat_grads(), self.loss_mean, self.group)        # RCCL over xGMI
            if loss_out is not None:
                loss_out.copy_(self.loss_mean, non_blocking=True)
        elif local and self.world > 1:
            keep, self.inv_batch = self.inv_batch, 1.0 / self.B
            try:
                self._forward_loss_backward(rows, loss_out, opt, update=True)
            finally:
                self.inv_batch = keep
        else:
            self._forward_loss_backward(rows, loss_out, opt, update=True)       # one rank: the optimiser rides in the backward
        if not self._updated:
            opt.apply(prepared=self._prepared, batch=self.B)

    def step(self, opt, rows_dev, loss_out=None):
        """One optimiser step on the int32 device index vector ``rows_dev[B]``; the mean loss of the step lands in
        ``loss_out`` (a 1-float device tensor) when given, in ``self.loss_mean`` otherwise."""
        if self.use_graph and self.graph is not None and self._graph_sig == (self.model.flat_params().data_ptr(), id(opt)):
            self.rows.copy_(rows_dev, non_blocking=True)
            _lib.call("linna_graph_launch", self.graph, _lib.stream())
            if loss_out is not None:
                loss_out.copy_(self.loss_mean, non_blocking=True)
        else:
            self._step_body(opt, rows_dev.contiguous(), loss_out)

    def prepare_graph(self, opt):
        """Capture one optimiser step.  Capture does not execute: parameters are untouched."""
        if not self.use_graph:
            return
        self.model.flat_grads()
        self.model.net_handle(with_grads=True)
        self.model.workspace(self.B)
        self.model.workspace(self.B, "bwd")
        if self.graph is not None:
            _lib.call("linna_graph_destroy", self.graph)
            self.graph = None
        # nothing runs before the capture: linna_net_prepare (below, and in __init__) has allocated the weight streams,
        # the descriptor table of the grouped parameter-gradient launch travels as kernel arguments
        _lib.call("linna_net_prepare", self.model.net_handle(with_grads=True), 1, 0)
        _lib.call("linna_net_prepare_loss", self.model.net_handle(with_grads=True), C.byref(self.desc))
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            st = _lib.stream()
            _lib.call("linna_graph_begin", st)
            self._step_body(opt)
            g = C.c_void_p()
            _lib.call("linna_graph_end", st, C.byref(g))
        torch.cuda.current_stream().wait_stream(side)
        self.graph = g
        self._graph_sig = (self.model.flat_params().data_ptr(), id(opt))

    # ------------------------------------------------------------------ validation (util.py:1124-1127)
    def validate_enqueue(self, model=None):
        """Launch the validation pass (util.py:1124-1127) without waiting for it: the epoch loop queues it right
        behind the epoch's optimiser steps and reads everything back with ONE synchronisation (three round trips
        per epoch left the GPU idle between them, and an idle MI355X drops its clocks)."""
        v, k, st = self.val, self.k, _lib.stream()
        n = v["n"]
        if "xb" not in v:
            z = lambda *s: torch.zeros(s, dtype=torch.float32, device=self.dev)
            v["xb"], v["pred"] = z(n, _lib.ld4(self.nin)), z(n, _lib.ld4(self.nout))
            v["scratch"] = z(_lib.load().linna_loss_scratch_bytes(n, self.nout) // 4 + 4)
            v["rows"] = z(2, n)                                  # loss_rows | frac_rows: one copy back
            v["loss_rows"], v["frac_rows"] = v["rows"][0], v["rows"][1]
        _lib.call("linna_gather_xform", self.ctx, _lib.ptr(v["X"]), v["X"].stride(0), None, n, self.nin,
                  _lib.iptr(k["lg"]) if k["lg"] is not None else None, _lib.ptr(k["xmean"]), _lib.ptr(k["xstd"]),
                  _lib.ptr(v["xb"]), v["xb"].stride(0), st)
        (model if model is not None else self.model).forward_buffer(v["xb"], n, out=v["pred"])      # (a shadow of the model: the epoch loop's snapshot)
        _lib.call("linna_val_rows", self.ctx, C.byref(self.desc), _lib.ptr(v["pred"]), v["pred"].stride(0), _lib.ptr(v["Y"]),
                  v["Y"].stride(0), _lib.ptr(v["den"]), n, _lib.ptr(v["scratch"]), _lib.ptr(v["loss_rows"]),
                  _lib.ptr(v["frac_rows"]), st)

    def metrics_enqueue(self, last_loss, out):
        """The epoch's record for the controller: out[4] (device) = last training loss, median validation loss, max and
        median of |chi2_nnd / chi2_Md - 1| (util.py:1124-1127) -- selected on the device (linna_val_metrics)."""
        v = self.val
        _lib.call("linna_val_metrics", self.ctx, _lib.ptr(v["loss_rows"]), _lib.ptr(v["frac_rows"]), v["n"],
                  _lib.ptr(last_loss) if last_loss is not None else None, _lib.ptr(out), _lib.stream())

    def validate_finish(self):
        out = torch.empty(4, dtype=torch.float32, device=self.dev)
        self.metrics_enqueue(None, out)
        return out[1:].cpu().numpy().astype(np.float64)

    def validate(self):
        self.validate_enqueue()
        return self.validate_finish()


def _read_lr(pred, engine, rank, size=1, group=None):
    """predictor_gpu.py:222-245: learning rate from lr.npy; rank 0 runs the range test if the file is absent -- on its
    own batches, without collectives (the other ranks are not in it) -- and every rank receives the value by broadcast
    (the reference's other ranks spin on the file)."""
    path = os.path.join(pred.outdir, "lr.npy") if pred.outdir is not None else None
    lr = 0.0
    if rank == 0:
        if path is not None and os.path.isfile(path):
            lr = float(np.load(path))
        else:
            from . import lrfinder
            lr = float(lrfinder.range_test(pred, engine))
            if path is not None:
                np.save(path + ".tmp.npy", lr)
                os.replace(path + ".tmp.npy", path)
    if size > 1:
        from . import dist as ldist
        lr = ldist.broadcast_value(lr, group, engine.dev)
    return lr


class _EpochProf(object):
    """Where the epochs of a training run spend their time (``Predictor.train(..., profile={})``): host seconds per phase
    (the marks partition the loop's wall time) and device seconds of the optimiser steps / the validation pass from event
    pairs on the launch stream; totals and per-epoch medians (a re-initialisation or a checkpoint write makes single
    epochs many times longer than the typical one)."""

    def __init__(self, out):
        self.out, self.t0, self.cur, self.epochs, self.ev = out, None, {}, [], []

    def mark(self, key=None):
        """Host time since the previous mark goes to ``key``."""
        if self.out is None:
            return
        t = time.perf_counter()
        if key is not None and self.t0 is not None:
            self.cur[key] = self.cur.get(key, 0.0) + t - self.t0
        self.t0 = t

    def event(self):
        if self.out is None:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def span(self, key, e0, e1):
        if self.out is not None:
            self.cur.setdefault("_ev", []).append((key, e0, e1))

    def end_epoch(self):
        if self.out is not None:
            self.epochs.append(self.cur)
            self.cur = {}

    def finish(self, **extra):
        if self.out is None:
            return
        torch.cuda.synchronize()
        rows = []
        for ep in self.epochs:
            r = {"host_" + k: v for k, v in ep.items() if k != "_ev"}
            for key, a, b in ep.get("_ev", []):
                r["gpu_" + key] = r.get("gpu_" + key, 0.0) + 1e-3 * a.elapsed_time(b)
            r["epoch"] = sum(v for k, v in ep.items() if k != "_ev")
            rows.append(r)
        keys = sorted({k for r in rows for k in r})
        for k in keys:
            col = np.array([r.get(k, 0.0) for r in rows])
            self.out[k + "_s"] = float(col.sum())
            self.out[k + "_median_s"] = float(np.median(col)) if len(col) else 0.0
        self.out.update(extra)


def run(pred, dataset, num_epochs, loss_fn, val_dataset, val_metric_fn, initfrombest, rank, size, dist_group,
        checkpoint_every, progress, patience=500, profile=None):
    """The body of ``Predictor.train``; returns (train_losses[steps], val_metrics[epochs, 3]).

    One host wait per epoch.  Everything the controller reads -- the last training loss and the three validation metrics,
    selected on the device -- arrives as ONE 4-float record in pinned memory, the per-step losses ride along in a second
    pinned buffer; while the GPU works through the epoch the host already draws the next epoch's sample order (torch's
    generator is put back first if the controller then re-initialises the weights, so the random stream is the reference's:
    train order, validation draw, [Xavier draws], next train order) and ships it, so the next epoch's first step is enqueued
    as soon as the record has been looked at."""
    t_run = time.perf_counter()
    prof = _EpochProf(profile)
    progress = progress or os.environ.get("LINNA_TRAIN_PROGRESS", "0") == "1"   # per-epoch train / validation loss
    torch.manual_seed(1234)                                                     # predictor_gpu.py:221
    size = max(int(size), 1)
    if size > 1:
        from . import dist as ldist
        ldist.init()        # rendezvous + the library's RCCL communicator (no-op when the launcher already did; predictor_gpu.py:240-252)
        ldist.enter("Predictor.train (data parallel, size = %d)" % size, dist_group)   # every rank must be here: fail within minutes, not never
    model = pred.model
    with _lib.stage("train_NN.engine_setup"):
        engine = TrainEngine(pred, dataset, loss_fn, val_dataset, world_size=size, dist_group=dist_group)
    if pred.optim == "automatic" or pred.optim is None:
        with _lib.stage("train_NN.lr_range_test"):
            lr = _read_lr(pred, engine, rank, size, dist_group)
    else:
        lr = float(getattr(pred.optim, "lr", 1e-3))
    lr = lr * size                                                              # :246
    pred.optim = None
    if initfrombest and pred.outdir is not None:
        if not pred.load_checkpoint():
            print("best.pth.tar does not exsit")
    opt = _AdamWState(model, lr, weight_decay=1e-4)                             # :267
    pred.optim = opt
    engine.prepare_graph(opt)
    es = EarlyStopping(patience=patience)                                       # :256 (500)
    ckpt = _Checkpoints(pred, model, rank)
    last_epoch = -1
    train_losses, val_metrics = [], []
    old, told = 0.0, 0.0
    best_state = None
    nsteps = len(dataset) // size               # every rank consumes its own batch of B rows per step
    loss_hist = torch.zeros(max(nsteps, 1), dtype=torch.float32, device=engine.dev)
    rec_dev = torch.zeros(4, dtype=torch.float32, device=engine.dev)
    rec_pin = torch.zeros(4, dtype=torch.float32).pin_memory()
    hist_pin = torch.zeros(max(nsteps, 1), dtype=torch.float32).pin_memory()
    rows_pin = [torch.zeros((max(nsteps, 1), engine.B), dtype=torch.int32).pin_memory() for _ in range(2)]
    rows_dev = [torch.zeros((max(nsteps, 1), engine.B), dtype=torch.int32, device=engine.dev) for _ in range(2)]
    pre = {"rows": None, "rng": None}            # the NEXT epoch's sample order, drawn ahead; torch's generator state before the draw

    def draw_rows(slot):
        """This rank's batches of one epoch on the device (same order on every rank -- same seed; step s of rank r takes
        global batch s * size + r, dist.rank_batches): one asynchronous copy from pinned memory."""
        if not nsteps:
            return None
        rows_pin[slot].numpy()[...] = dataset.epoch_rows()[rank::size][:nsteps]
        rows_dev[slot].copy_(rows_pin[slot], non_blocking=True)
        return rows_dev[slot]

    def drop_prefetch():
        """The order drawn ahead will not be used as drawn: torch's generator goes back to where it was before the draw."""
        if pre["rng"] is not None:
            torch.set_rng_state(pre["rng"])
            torch.cuda.current_stream().synchronize()      # (its copy from the pinned buffer is done before that is rewritten)
        pre["rows"], pre["rng"] = None, None

    # -- speculation: the next epoch's steps start BEFORE this epoch's verdict ---------------------------------------------
    # The controller needs the validation metrics of epoch i, and nearly always answers "carry on".  One rank: epoch i ends
    # with a device copy of (parameters, m, v, step) into a shadow model; the validation pass of epoch i runs on that shadow
    # on a stream of its own WHILE the launch stream already works through the steps of epoch i + 1.  When the verdict does
    # change something -- learning rate / weight decay, a re-initialisation, the best weights restored, a stop -- the live
    # state is first put back from the shadow (in stream order behind the speculative steps) and epoch i + 1 is enqueued
    # again: the trajectory is the sequential one, bit for bit.  Checkpoints and the best state are taken from the shadow.
    spec = (size == 1 and val_dataset is not None and nsteps > 0 and os.environ.get("LINNA_TRAIN_SPECULATE", "1") != "0")
    shadow, snap, side = model, None, None
    if spec:
        import copy
        shadow = copy.deepcopy(model)                 # same topology, a flat parameter buffer and weight streams of its own
        snap = dict(m=torch.zeros_like(opt.m), v=torch.zeros_like(opt.v), step=torch.zeros_like(opt.step_dev))
        side = torch.cuda.Stream(device=engine.dev)
    last_dev = torch.zeros(1, dtype=torch.float32, device=engine.dev)
    fly = {"on": False, "epochs": 0, "undone": 0, "quiet": 0, "acted": False, "acts": {}}   # on: a speculative epoch is enqueued on top of the state the verdict is about
    QUIET = 4           # speculate only after so many consecutive "carry on" verdicts: a controller that is busy (a run
                        # that keeps restoring its best weights, say) then costs no wasted epochs, a quiet one loses none

    def state_of_epoch():
        """(parameters, m, v, step) as they were at the end of the epoch the controller is looking at."""
        if spec:
            return shadow._flat, snap["m"], snap["v"], snap["step"]
        return model._flat, opt.m, opt.v, opt.step_dev

    def rollback(why=None):
        """Before the controller changes anything: undo the speculative steps (stream-ordered copies, no host wait)."""
        fly["acted"] = True
        if why is not None:
            fly["acts"][why] = fly["acts"].get(why, 0) + 1
        if fly["on"]:
            model._flat.copy_(shadow._flat)
            opt.m.copy_(snap["m"]); opt.v.copy_(snap["v"]); opt.step_dev.copy_(snap["step"])
            model.weights_changed()                   # the training streams are re-laid from the restored parameters
            fly["on"] = False
            fly["undone"] += 1

    def new_optimizer(lr_now):
        nonlocal opt
        rollback()
        opt = _AdamWState(model, lr_now, weight_decay=1e-4)
        pred.optim = opt
        engine.prepare_graph(opt)

    def halve_lr():
        if opt.lr > 2e-6:
            rollback()
            print("learning rate too large: {0}".format(opt.lr), flush=True)
            opt.lr = opt.lr / 2.0
            opt.push_hyper()

    def reinit():
        rollback()
        drop_prefetch()                           # the Xavier draws come BEFORE the next epoch's order in the reference
        model.init_weight()                       # fresh Xavier weights in place (model_old.init_weight(), :323)

    def enqueue_steps(perm):
        e0 = prof.event()
        for s_ in range(nsteps):
            engine.step(opt, perm[s_], loss_hist[s_:s_ + 1])                    # :273-288
        prof.span("steps", e0, prof.event())

    def enqueue_tail():
        """Behind an epoch's steps: its losses to pinned memory, the state the verdict will be about, the validation pass
        and the controller's record.  Returns the event behind which the record is in pinned memory."""
        hist_pin.copy_(loss_hist, non_blocking=True)
        if val_dataset is None:
            ev = torch.cuda.Event(); ev.record()
            return ev
        val_dataset.epoch_batches()                                             # keeps torch's RNG stream aligned
        if nsteps:
            last_dev.copy_(loss_hist[nsteps - 1:nsteps])
        e1 = prof.event()
        if not spec:
            engine.validate_enqueue()                                           # queued behind the steps: one wait per epoch
            engine.metrics_enqueue(last_dev if nsteps else None, rec_dev)
            rec_pin.copy_(rec_dev, non_blocking=True)
            prof.span("validation", e1, prof.event())
            ev = torch.cuda.Event(); ev.record()
            return ev
        shadow._flat.copy_(model._flat)
        snap["m"].copy_(opt.m); snap["v"].copy_(opt.v); snap["step"].copy_(opt.step_dev)
        ready = torch.cuda.Event(); ready.record()
        with torch.cuda.stream(side):
            side.wait_event(ready)
            engine.validate_enqueue(shadow)
            engine.metrics_enqueue(last_dev, rec_dev)
            rec_pin.copy_(rec_dev, non_blocking=True)
            ev = torch.cuda.Event(); ev.record(side)
        return ev

    perm = draw_rows(0)
    i = 0
    prof.mark()
    if num_epochs > 0:
        enqueue_steps(perm)
    prof.mark("enqueue_steps")
    with _lib.stage("train_NN.epochs"), _lib.quiet_gc():   # (the epochs' launches are queued 2-9 ms ahead: no full collector pass inside the loop;
                                             #  a context manager: an exception inside the loop hands the objects back to the collector)
        while i < num_epochs:
            landed = enqueue_tail()
            prof.mark("enqueue_validation")
            if i + 1 < num_epochs and nsteps:                                       # the next epoch's order, while the GPU works
                pre["rng"] = torch.get_rng_state()
                pre["rows"] = draw_rows((i + 1) & 1)
            prof.mark("rows")
            if spec and pre["rows"] is not None and fly["quiet"] >= QUIET:
                fly["on"] = True
                fly["epochs"] += 1
                enqueue_steps(pre["rows"])                                          # epoch i + 1, on the assumption "carry on"
                prof.mark("enqueue_steps")
            landed.synchronize()                                                    # THE wait of the epoch
            prof.mark("wait")
            epoch_losses = hist_pin.numpy()[:nsteps].astype(np.float64)
            train_losses.extend(epoch_losses.tolist())
            loss = float(epoch_losses[-1]) if nsteps else float("nan")
            is_best = False
            stop = False
            if val_dataset is not None:
                vm = rec_pin.numpy()[1:].astype(np.float64)
                val_metrics.append(vm)
                if progress and rank == 0:
                    print("epoch %d  train %.5e  val %.5e" % (i, loss, vm[0]), flush=True)
                if pred.outdir is not None:
                    is_best = vm[0] < pred.best_val_loss
                    if is_best:
                        pred.best_val_loss = vm[0]
                recent = np.array(val_metrics[-10:])[:, 0]
                if np.std(recent) < 0.01 * np.mean(recent) and 10 <= i < 120 and i % 10 == 0:      # :319-335
                    print("bad trainning: {0}".format(i), flush=True)
                    lr_now = opt.lr
                    rollback("plateau: re-initialised")
                    reinit()
                    new_optimizer(lr_now)
                    if i > 10 and lr_now > 2e-4:
                        halve_lr()
                v0 = val_metrics[-1][0]
                if np.isnan(v0) or v0 > 1e10 or (v0 - old > 5 * old and i != 0) or (loss - told > 5 * told and i != 0):  # :339
                    lr_now = opt.lr
                    restored = False
                    rollback("loss jump / NaN: best weights restored")
                    if best_state is not None:
                        model.flat_params().copy_(best_state)
                        restored = True
                    elif pred.outdir is not None:
                        drain_checkpoints()                                         # (no best yet in this run: an older file)
                        restored = pred.load_checkpoint(ismpi=False)
                    if not restored:
                        reinit()
                    new_optimizer(lr_now)
                    if np.isnan(v0) or v0 > 1e10 or (v0 - old > 10 * old):
                        if i > 10:
                            halve_lr()
                    if not np.isnan(v0) and (v0 - old > 5 * old):
                        val_metrics[-1][0] = old
                else:
                    criteria = es.step(v0, loss)                                    # :375-401
                    if criteria == 1:
                        if opt.lr > 2e-6:
                            rollback("early stopping: lr / 2")
                            print("\n learning rate too large: {0}\n".format(opt.lr), flush=True)
                            opt.lr, opt.weight_decay = opt.lr / 2.0, opt.weight_decay / 2
                            opt.push_hyper()
                        else:
                            es.cooling = 0
                    if criteria == 2:
                        print("early stop", flush=True)
                        print("learning rate", opt.lr, flush=True)
                        # the reference breaks on rank 0 only (predictor_gpu.py:392-393), harmless there because nothing
                        # collective follows; here every rank holds the same metrics (all-reduced loss, replicated
                        # validation set) and the next epoch starts with an all-reduce, so every rank stops
                        rollback("early stopping: stop")
                        drop_prefetch()
                        stop = True
                    if criteria == 3:
                        print("\n weight decay too small: {0}\n".format(opt.weight_decay), flush=True)
                        if opt.weight_decay < 1e0:
                            rollback("early stopping: weight decay x 2")
                            opt.weight_decay = opt.weight_decay * 2
                            opt.push_hyper()
                old = val_metrics[-1][0]
                told = loss
            prof.mark("controller")
            p_, m_, v_, st_ = state_of_epoch() if fly["on"] else (model._flat, opt.m, opt.v, opt.step_dev)
            if is_best:
                best_state = p_.clone()                                             # device-resident best.pth.tar
            ckpt.record(opt, i, is_best, checkpoint_every, num_epochs, force=stop, state=(p_, m_, v_, st_))
            prof.mark("checkpoint")
            prof.end_epoch()
            last_epoch = i
            if stop:
                break
            fly["quiet"] = 0 if fly["acted"] else fly["quiet"] + 1
            fly["acted"] = False
            i += 1
            if i < num_epochs and not fly["on"]:                                    # not speculated, or undone: (re)enqueue the epoch
                if pre["rows"] is not None:
                    perm, pre["rows"], pre["rng"] = pre["rows"], None, None
                else:
                    perm = draw_rows(i & 1)
                prof.mark("rows")
                enqueue_steps(perm)
                prof.mark("enqueue_steps")
            else:
                pre["rows"], pre["rng"] = None, None                                # (the speculative epoch has consumed the order drawn ahead)
            fly["on"] = False
    t_loop = time.perf_counter()
    with _lib.stage("train_NN.final_checkpoint"):
        ckpt.finish(opt, last_epoch)                # best.pth.tar / last.pth.tar are on disk when train() returns
    prof.finish(epochs=last_epoch + 1, steps_per_epoch=nsteps, total_s=time.perf_counter() - t_run,
                final_checkpoint_s=time.perf_counter() - t_loop, speculative_epochs=fly["epochs"], speculative_epochs_undone=fly["undone"], controller_actions=dict(fly["acts"]))
    if val_dataset is not None:
        return np.array(train_losses), np.array(val_metrics)
    return np.array(train_losses)


class _CheckpointWriter(object):
    """last.pth.tar / best.pth.tar written by ONE background thread, in order: pickling and writing
    ~10 MB took 17 ms of every 25-50 ms epoch on the training thread (the reference pays the same
    per epoch, predictor_gpu.py:405-419).  The state handed over is already on the host; ``drain``
    returns when every file is on disk and is called before anything reads a checkpoint back."""

    def __init__(self):
        import queue, threading
        self.q = queue.Queue()
        self.err = None
        self.t = threading.Thread(target=self._work, daemon=True)
        self.t.start()

    def _work(self):
        while True:
            item = self.q.get()
            try:
                if item is not None and self.err is None:
                    if item[0] == "file":
                        nnutils.save_state(item[1], item[2])
                    else:
                        nnutils.save_checkpoint(*item)
            except Exception as e:                      # surfaced by drain()
                self.err = e
            finally:
                self.q.task_done()
            if item is None:
                return

    def put(self, state, is_best, checkpoint):
        self.q.put((state, is_best, checkpoint))

    def put_file(self, state, path):
        self.q.put(("file", state, path))

    def drain(self):
        self.q.join()
        if self.err is not None:
            err, self.err = self.err, None
            raise err


_writer = None


def drain_checkpoints():
    if _writer is not None:
        _writer.drain()


class _Checkpoints(object):
    """predictor_gpu.py:405-419 keeps last.pth.tar current every epoch and copies it to best.pth.tar on
    improvement.  Here every epoch only RECORDS: the best state is a device-side copy (parameters, AdamW m / v,
    step), and the files are (re)written at most every ``interval`` seconds and when training ends, by the
    background writer -- pickling ~10 MB per epoch cost a third of the wall time of a training run.  After
    ``train()`` returns the two files hold exactly what the reference's would (last epoch, best epoch);
    a crash loses at most ``interval`` seconds (LINNA_CHECKPOINT_INTERVAL, 0 = write every epoch)."""

    def __init__(self, pred, model, rank):
        self.pred, self.model = pred, model
        self.on = pred.outdir is not None and rank == 0
        self.interval = float(os.environ.get("LINNA_CHECKPOINT_INTERVAL", "2.0"))
        self.t_last = time.time()
        self.best, self.best_dirty, self.last_epoch_written = None, False, -1

    def record(self, opt, epoch, is_best, every, num_epochs, force=False, state=None):
        """``state`` = (parameters, m, v, step) of the epoch being recorded when that is not the live state (the epoch
        loop runs one epoch ahead of its verdicts: trainer.run)."""
        if not self.on:
            return
        p, m, v, step = state if state is not None else (self.model._flat, opt.m, opt.v, opt.step_dev)
        if is_best:
            self.best = dict(epoch=epoch, p=p.detach().clone(), opt=(m.clone(), v.clone(), step.clone(), opt.lr, opt.weight_decay))
            self.best_dirty = True
        due = is_best or force or (epoch + 1) % max(every, 1) == 0 or epoch + 1 == num_epochs
        if force or epoch + 1 == num_epochs or (due and time.time() - self.t_last >= self.interval):
            self.write(opt, epoch, (p, m, v, step))

    def _host_state(self, flat, epoch, optim_dict):
        host = flat.detach().cpu()
        sd = {k: self.model._view(host, k).clone().contiguous() for k in self.model._index}
        return {"epoch": epoch + 1, "state_dict": sd, "optim_dict": optim_dict}

    def write(self, opt, epoch, state=None):
        global _writer
        if _writer is None:
            _writer = _CheckpointWriter()
        out = self.pred.outdir
        p, m, v, step = state if state is not None else (self.model._flat, opt.m, opt.v, opt.step_dev)
        last = self._host_state(p, epoch, opt.state_dict(snapshot=(m, v, step, opt.lr, opt.weight_decay)))
        if self.best_dirty and self.best["epoch"] == epoch:
            _writer.put(last, True, out)                                       # last.pth.tar + copy to best.pth.tar
        else:
            _writer.put(last, False, out)
            if self.best_dirty:
                b = self.best
                _writer.put_file(self._host_state(b["p"], b["epoch"], opt.state_dict(snapshot=b["opt"])),
                                 os.path.join(out, "best.pth.tar"))
        self.best_dirty = False
        self.t_last = time.time()
        self.last_epoch_written = epoch

    def finish(self, opt, epoch):
        if self.on and (self.best_dirty or self.last_epoch_written != epoch) and epoch >= 0:
            self.write(opt, epoch)
        drain_checkpoints()



