"""Is the training step host-bound?  Time to ENQUEUE n steps (no sync) against time to finish them."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.argv = sys.argv[:1]
import numpy as np, torch, bench_paths
from bench_paths import *
p = problem("ChtoModelv2", 33, 33, True)
rs = np.random.RandomState(3); n = 20000; B = 500
X = (p["X_mean"][None, :] + p["X_std"][None, :] * rs.standard_normal((n, 33))).astype(np.float32)
Y = (p["data"][None, :] + 3 * p["sigma"][None, :] * rs.standard_normal((n, 33))).astype(np.float32)
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
for n in (20, 100, 400):
    t0 = time.perf_counter()
    for i in range(n):
        eng.step(opt, perm[i % len(perm)])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("steps %4d: enqueue %.1f us/step, finished after %.1f us/step" % (n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
