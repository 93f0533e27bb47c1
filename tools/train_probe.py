"""Profile target: 100 training steps of ChtoModelv2(33,33), batch 500 (direct launches)."""
import sys, os
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
for i in range(120):
    eng.step(opt, perm[i % len(perm)])
torch.cuda.synchronize()
