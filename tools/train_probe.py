"""Profile target: training steps at batch 500 (direct launches).  Usage: train_probe.py [nin nout [nsteps [B]]]
(default 26 457: the bench's `training` workload, dense covariance)."""
import sys, os
args = sys.argv[1:]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.argv = sys.argv[:1]
import numpy as np, torch, bench_paths, time
from bench_paths import *
nin, nout = (int(args[0]), int(args[1])) if len(args) >= 2 else (26, 457)
nsteps = int(args[2]) if len(args) >= 3 else 200
B = int(args[3]) if len(args) >= 4 else 500
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
for i in range(300):
    eng.step(opt, perm[i % len(perm)])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(nsteps):
    eng.step(opt, perm[i % len(perm)])
torch.cuda.synchronize()
print("ChtoModelv2(%d,%d) batch %d: %.1f us per step" % (nin, nout, B, 1e6 * (time.perf_counter() - t0) / nsteps))
