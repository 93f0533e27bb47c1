#!/usr/bin/env python
"""Diagnostic: phase stamps of the two whole-network launches of a training step (library built with
LINNA_HIPCC_EXTRA=-DNS_STAMPS).  Usage: python tools/ns_stamps_train.py [nin nout [B]]
Stamp order: start, after the prologue barrier, after every segment's last barrier, after the loop, end."""
import os, sys
args = sys.argv[1:]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.argv = sys.argv[:1]
import numpy as np, torch
nin, nout = (int(args[0]), int(args[1])) if len(args) >= 2 else (26, 457)
B = int(args[2]) if len(args) >= 3 else 500
nb = (B + 3) // 4
buf = torch.zeros(nb * 8 * 32, dtype=torch.int64, device="cuda")
os.environ["LINNA_FUSED_STAMPS"] = "%x" % buf.data_ptr()
import bench_paths
from bench_paths import *
from linna_amd import _lib
p = problem("ChtoModelv2", nin, nout, True)
rs = np.random.RandomState(3); n = 20000
X = (p["X_mean"][None, :] + p["X_std"][None, :] * rs.standard_normal((n, nin))).astype(np.float32)
Y = (p["data"][None, :] + 3 * p["sigma"][None, :] * rs.standard_normal((n, nout))).astype(np.float32)
ytd = util.Y_transform_data(p["sigma"], "cpu")
yinv = util.Y_invtransform_class(t32(p["y_mean"]), t32(p["y_std"]), t32(p["data"]), "cpu")
lf = util.Loss_fn(t32(p["data"]), torch.tensor(p["cov"], dtype=torch.float64),
                  torch.tensor(np.linalg.inv(p["cov"]), dtype=torch.float64), ytd, yinv, "cpu")
loader = predictor_gpu.BatchLoader(util.ArrayDataset(X, Y), B, shuffle=True, drop_last=True)
eng = trainer.TrainEngine(p["pred"], loader, lf, None, use_graph=False)
opt = predictor_gpu._AdamWState(p["model"], 1e-4)
perm = torch.stack(loader.epoch_batches()).to(torch.int32).cuda()


def report(tag):
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(nb, 8, 32).astype(np.float64)
    n = int((t[0, 0] > 0).sum())
    d = np.diff(t[:, :, :n], axis=2)
    print("%s: %d stamps" % (tag, n))
    print("phase   median cycles   (max over waves, median over blocks)")
    for i in range(n - 1):
        print("%2d -> %2d %10.0f %10.0f" % (i, i + 1, np.median(d[:, :, i]), np.median(d[:, :, i].max(1))))
    print("total per wave median %.0f cycles; first start -> last end over the grid %.0f" % (
        np.median(t[:, :, n - 1] - t[:, :, 0]), t[:, :, n - 1].max() - t[:, :, 0].min()), flush=True)


import ctypes as C
for i in range(20):
    eng.step(opt, perm[i % len(perm)])
torch.cuda.synchronize()
# the dX chain is the last whole-network launch of a step
report("dX chain (STORE == 2)")
# the forward + loss launch alone, through its own entry
m, k, rows = p["model"], eng.k, perm[0]
buf.zero_()
_lib.call("linna_net_forward_loss", m.net_handle(with_grads=True), C.byref(eng.desc), _lib.ptr(eng.X), eng.X.stride(0),
          _lib.iptr(rows), B, _lib.iptr(k["lg"]) if k["lg"] is not None else None, _lib.ptr(k["xmean"]), _lib.ptr(k["xstd"]),
          _lib.ptr(eng.xb), eng.xb.stride(0), _lib.ptr(m.workspace(B)), _lib.ptr(eng.predb), eng.predb.stride(0),
          _lib.ptr(eng._targets()), eng.YN.stride(0), _lib.ptr(eng.den), eng.inv_batch, _lib.ptr(eng.loss_rows),
          _lib.ptr(eng.loss_mean), _lib.ptr(eng.dpred), eng.dpred.stride(0), None, None, 0.0, 0.0, _lib.stream())
report("forward + loss (STORE == 3)")
