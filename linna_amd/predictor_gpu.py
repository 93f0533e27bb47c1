"""Emulator training and prediction on MI355X.

Mirrors ``linna/predictor_gpu.py``: ``EarlyStopping`` (:19-150, same return codes),
``Predictor(in_size, out_size, model, optim, X_transform, y_transform, device, scheduler,
outdir)`` with ``train`` (:201-449), ``predict`` (:461-504), ``load_checkpoint`` (:451-459).

The minibatch step -- gather + input transform, forward, chi^2-ratio loss, backward, AdamW
-- is a chain of HIP kernels over device-resident data, launched directly (a hipGraph replay of
the same step is available, ``TrainEngine(use_graph=True)``: measured equal or slower on ROCm 7.2);
only the epoch-level controller (learning-rate / weight-decay schedule, divergence recovery,
checkpoints) runs on the host, with the reference's semantics.
"""
import copy
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from . import nnutils
from .nn import *  # noqa: F401,F403


def _lower_median(values):
    """torch.median semantics (lower middle element)."""
    s = sorted(values)
    return s[(len(s) - 1) // 2]


class EarlyStopping(object):
    """Learning-rate / weight-decay / stop controller (predictor_gpu.py:19-150).

    ``step(val, train)`` returns 0 (continue), 1 (halve lr and weight decay), 2 (stop),
    3 (double weight decay)."""

    def __init__(self, mode="min", min_delta=0, patience=10, nqueue=200, percentage=False):
        if mode not in ("min", "max"):
            raise ValueError("mode " + mode + " is unknown!")
        self.mode, self.min_delta, self.patience, self.nqueue = mode, min_delta, patience, nqueue
        self.percentage = percentage
        self.best = None
        self.best_t = None
        self.num_bad_epochs = 0
        self.cooling = 0
        self.cooling_weight_decay = 0
        self.queue_t, self.queue_v = [], []

    def is_better(self, a, best):
        if self.patience == 0:
            return True
        d = best * self.min_delta / 100 if self.percentage else self.min_delta
        return a < best - d if self.mode == "min" else a > best + d

    def step(self, metrics, metrics_t):
        if self.patience == 0:
            return False
        metrics_t = float(metrics_t)
        self.queue_t.append(metrics_t)
        self.queue_v.append(metrics)
        if len(self.queue_t) > self.nqueue:
            self.queue_t.pop(0)
        if len(self.queue_v) > self.nqueue:
            self.queue_v.pop(0)
        trend = None
        if len(self.queue_t) > 2:
            ht, hv = int(0.5 * len(self.queue_t)), int(0.5 * len(self.queue_v))
            trend = (_lower_median(self.queue_t[ht:]) - _lower_median(self.queue_t[:ht]),
                     float(np.median(self.queue_v[hv:])) - float(np.median(self.queue_v[:hv])))
        if self.best is None:
            self.best, self.best_t, self.num_bad_epochs = metrics, metrics_t, 0
            return 0
        if np.isnan(metrics):
            print("nan metric", flush=True)
            self.num_bad_epochs += 1
            return 0
        if self.is_better(metrics, self.best):
            self.num_bad_epochs = 0
            self.cooling = 0
            self.cooling_weight_decay = 0
            self.best, self.best_t = metrics, metrics_t
        else:
            self.num_bad_epochs += 1
            if 0.9 * self.patience <= self.num_bad_epochs < self.patience:
                if self.cooling != 0:
                    if self.cooling > 500:
                        self.cooling = 0
                        self.num_bad_epochs += 5
                    else:
                        self.num_bad_epochs -= 1
                        self.cooling += 1
                    return 0
                self.cooling += 1
                return 1
            if trend is not None and len(self.queue_t) > 0.5 * self.nqueue and trend[0] < 0 and trend[1] > 0:
                # training loss still falling while validation rises: over-fitting
                if self.cooling_weight_decay != 0:
                    if self.cooling_weight_decay > 1000:
                        self.cooling_weight_decay = 0
                        return 0
                    self.queue_t, self.queue_v = [], []
                    self.cooling_weight_decay += 1
                    return 3 if self.cooling_weight_decay % 50 == 0 else 0
                self.cooling_weight_decay += 1
                return 3
        if self.num_bad_epochs >= self.patience:
            return 2
        return 0


class BatchLoader(object):
    """Index-only stand-in for the ``torch.utils.data.DataLoader`` the reference builds at
    util.py:1285-1286.  The arrays stay resident on the GPU; an epoch is a list of index batches.
    ``epoch_batches`` consumes torch's global RNG exactly as iterating the DataLoader does (one
    draw for the iterator's base seed, one for the RandomSampler seed, then ``randperm`` on a
    private generator), so with ``torch.manual_seed(1234)`` (predictor_gpu.py:221) the sample
    order matches the reference's."""

    def __init__(self, dataset, batch_size, shuffle=False, drop_last=False):
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, int(batch_size), shuffle, drop_last

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def epoch_order(self):
        """This epoch's sample order (int64 tensor [n]); consumes torch's global RNG as iterating the DataLoader does."""
        n = len(self.dataset)
        torch.empty((), dtype=torch.int64).random_()                     # _BaseDataLoaderIter base seed
        if not self.shuffle:
            return torch.arange(n)
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        g = torch.Generator()
        g.manual_seed(seed)
        # ONE thread.  Above 32768 elements torch's CPU ops go through its intra-op thread pool, sized by the host's core
        # count (256 on an MI355X box) whatever the CPU quota of the container (16): a parallel region then burns the
        # cgroup's quota in a third of the scheduling period and the WHOLE process is frozen for the rest of it -- every
        # epoch of more than 64 steps (66 x 500 rows) took 40 ms instead of 14 (and randperm(40000) itself 2-47 ms).
        # The permutation is the same.
        nt = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            return torch.randperm(n, generator=g)
        finally:
            torch.set_num_threads(nt)

    def epoch_rows(self):
        """The epoch's batches as ONE int32 numpy array [batches, batch_size] (numpy: never multi-threaded, see
        ``epoch_order``); the training loop's form of ``epoch_batches``."""
        order = self.epoch_order().numpy()
        nb, B = len(self), self.batch_size
        if nb * B <= len(order):
            return np.ascontiguousarray(order[:nb * B].reshape(nb, B), dtype=np.int32)
        raise ValueError("epoch_rows needs equal batches (drop_last=True or a batch size dividing the set)")

    def epoch_batches(self):
        order = self.epoch_order()
        nb = len(self)
        return [order[i * self.batch_size:(i + 1) * self.batch_size] for i in range(nb)]


class _AdamWState(object):
    """AdamW hyper-parameters + moments over the model's flat buffer (device resident);
    (de)serialises to the ``torch.optim.AdamW.state_dict()`` layout the reference stores."""

    def __init__(self, model, lr, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-8):
        flat = model.flat_params()
        self.model = model
        self.lr, self.weight_decay, self.betas, self.eps = float(lr), float(weight_decay), betas, eps
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=flat.device)
        self.hyper = torch.zeros(4, dtype=torch.float32, device=flat.device)
        self._streams = None         # None: try linna_net_adamw_step when apply() is told the batch size
        self.push_hyper()

    @property
    def param_groups(self):
        return [self]

    def __getitem__(self, k):       # param_group['lr'] style access
        return getattr(self, k)

    def push_hyper(self):
        self.hyper[:2].copy_(torch.tensor([self.lr, self.weight_decay], dtype=torch.float32))

    def apply(self, prepared=False, batch=None):
        """``prepared``: this step's counter / bias corrections were advanced by ``linna_net_forward_loss`` already.
        ``batch``: the batch size of the training steps around this update -- the update then also writes the two weight
        streams those steps read (``linna_net_adamw_step``, one launch for three) where the network trains through them."""
        flat = self.model._flat          # (not flat_params(): the entries below announce the change themselves)
        n = flat.numel()
        if batch is not None and self._streams is not False:
            rc = _lib.load().linna_net_adamw_step(
                self.model.net_handle(with_grads=True), int(batch), _lib.ptr(flat),
                _lib.ptr(self.model.flat_grads()), _lib.ptr(self.m), _lib.ptr(self.v), n, _lib.ptr(self.hyper),
                _lib.iptr(self.step_dev), self.betas[0], self.betas[1], self.eps, 1 if prepared else 0, _lib.stream())
            if rc == 0:
                self._streams = True
                return
            if rc != _lib.ERR_UNSUPPORTED or self._streams is True:
                _lib.check(rc)
            self._streams = False
        _lib.call("linna_adamw_step", _lib.ctx(self.m.device.index), _lib.ptr(flat),
                  _lib.ptr(self.model.flat_grads()), _lib.ptr(self.m), _lib.ptr(self.v), n, _lib.ptr(self.hyper),
                  _lib.iptr(self.step_dev), self.betas[0], self.betas[1], self.eps, 1 if prepared else 0, _lib.stream())

    def state_dict(self, snapshot=None):
        """torch.optim.AdamW-shaped state; ``snapshot`` = (m, v, step_dev, lr, weight_decay) device copies taken at
        another time (the best epoch) instead of the live state."""
        m, v, step_dev, lr, wd = snapshot if snapshot is not None else (self.m, self.v, self.step_dev, self.lr, self.weight_decay)
        step = float(step_dev.item())
        m_host, v_host = m.detach().cpu(), v.detach().cpu()                # one copy each, then host views
        state = {}
        for i, key in enumerate(self.model._index):
            state[i] = {"step": torch.tensor(step), "exp_avg": self.model._view(m_host, key).clone(),
                        "exp_avg_sq": self.model._view(v_host, key).clone()}
        group = {"lr": lr, "betas": self.betas, "eps": self.eps, "weight_decay": wd,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "params": list(range(len(self.model._index)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        groups = sd.get("param_groups", [])
        if groups:
            self.lr = float(groups[0].get("lr", self.lr))
            self.weight_decay = float(groups[0].get("weight_decay", self.weight_decay))
        st = sd.get("state", {})
        step = 0
        for i, key in enumerate(self.model._index):
            if i in st:
                self.model._view(self.m, key).copy_(torch.as_tensor(st[i]["exp_avg"]))
                self.model._view(self.v, key).copy_(torch.as_tensor(st[i]["exp_avg_sq"]))
                step = int(float(st[i]["step"]))
        self.step_dev.fill_(step)
        self.push_hyper()


class Predictor(object):
    """Training and batched evaluation of the emulator (predictor_gpu.py:153-504)."""

    def __init__(self, in_size=None, out_size=None, model=None, optim=None, X_transform=None, y_transform=None,
                 device="cpu", scheduler=None, outdir=None):
        self.in_size, self.out_size = in_size, out_size
        self.device = device
        self.best_val_loss = float("inf")
        self.outdir = outdir
        if model is None:
            model = ChtoModelv2(in_size, out_size, None)      # noqa: F405
        self.model = model.to(device)
        self.scheduler = scheduler
        self.optim = optim
        self.X_transform = X_transform
        self.y_transform = y_transform
        self.MKLDNN = False          # attributes the reference's main.py sets (main.py:266-268); no-ops here
        self.MKLDNNMODEL = False
        self._consts = None

    # ------------------------------------------------------------------ constants on the device
    def _device_consts(self):
        dev = self.model.device
        if self._consts is not None and self._consts["dev"] == dev:
            return self._consts
        nin, nout = self.model.in_size, self.model.out_size
        f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32), device=dev)
        Xt, Yt = self.X_transform, self.y_transform
        k = {"dev": dev}
        lg = np.zeros(nin, np.int32)
        if Xt is not None and not callable(getattr(Xt, "X_mean", None)) and hasattr(Xt, "X_mean"):
            k["xmean"], k["xstd"] = f32(_t2n(Xt.X_mean)), f32(_t2n(Xt.X_std))
            if getattr(Xt, "dolog10index", None) is not None:
                lg[list(Xt.dolog10index)] = 1
        elif Xt is None:
            k["xmean"], k["xstd"] = f32(np.zeros(nin)), f32(np.ones(nin))
        else:
            raise NotImplementedError("X_transform must be an X_transform_class (constants are fused into the kernels)")
        k["lg"] = torch.as_tensor(lg, device=dev) if lg.any() else None
        cm = _lib.ColMap()
        if Yt is not None and hasattr(Yt, "y_mean"):
            k["ymean"], k["ystd"] = f32(_t2n(Yt.y_mean)), f32(_t2n(Yt.y_std))
            cm.cscale, cm.cshift = _lib.ptr(k["ystd"]), _lib.ptr(k["ymean"])
            cm.cexp = 1 if getattr(Yt, "ypositive", False) else 0
        elif Yt is not None:
            raise NotImplementedError("y_transform must be a Y_transform_class")
        k["colmap"] = cm if Yt is not None else None
        self._consts = k
        return k

    # ------------------------------------------------------------------ prediction
    def predict(self, X, no_grad=True):
        """y_transform(model(X_transform(X))) for ``X[B, nin]`` or ``X[nin]`` (predictor_gpu.py:461-504)."""
        self.model.eval()
        if not torch.is_tensor(X):
            X = torch.as_tensor(np.asarray(X, np.float32))
        one = X.dim() == 1
        Xb = X.view(1, -1) if one else X
        k = self._device_consts()
        dev = k["dev"]
        Xb = Xb.detach().to(device=dev, dtype=torch.float32).contiguous()
        B, nin = Xb.shape
        xn = torch.empty((B, _lib.ld4(nin)), dtype=torch.float32, device=dev)
        _lib.call("linna_gather_xform", _lib.ctx(dev.index), _lib.ptr(Xb), Xb.stride(0), None, B, nin,
                  _lib.iptr(k["lg"]) if k["lg"] is not None else None, _lib.ptr(k["xmean"]), _lib.ptr(k["xstd"]),
                  _lib.ptr(xn), xn.stride(0), _lib.stream())
        y = self.model.forward_buffer(xn, B, colmap=k["colmap"])
        return y.reshape(-1) if one else y

    def load_checkpoint(self, ismpi=False):
        """predictor_gpu.py:451-459."""
        path = os.path.join(self.outdir, "best.pth.tar")
        if os.path.isfile(path):
            opt = self.optim if isinstance(self.optim, _AdamWState) else None
            nnutils.load_checkpoint(path, self.model, opt, device=self.device, ismpi=ismpi)
            return True
        return False

    # ------------------------------------------------------------------ training
    def train(self, dataset, num_epochs, loss_fn, val_dataset=None, val_metric_fn=None, initfrombest=False, pool=None,
              nocpu=False, rank=0, size=1, dist_group=None, checkpoint_every=1, progress=False, patience=500, profile=None):
        """predictor_gpu.py:201-449.  ``dist_group``, ``checkpoint_every``, ``progress``, ``patience`` (the
        reference hard-wires 500, :256) and ``profile`` (a dict that receives where the epochs' time went) are additions
        with the reference's behaviour as default."""
        from . import trainer          # the HIP training engine (kept separate from the API shell)
        return trainer.run(self, dataset, num_epochs, loss_fn, val_dataset, val_metric_fn, initfrombest, rank, size,
                           dist_group, checkpoint_every, progress, patience, profile)


def _t2n(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
