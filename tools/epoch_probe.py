"""Per-epoch overhead of Predictor.train: wall time of N epochs against steps x (time per step)."""
import sys, os, time, tempfile
_args = sys.argv[1:]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.argv = sys.argv[:1]
import numpy as np, torch, bench_paths
from bench_paths import *
p = problem("ChtoModelv2", 33, 33, True)
rs = np.random.RandomState(3); n = int(_args[0]) if _args else 10000; nv = int(_args[1]) if len(_args) > 1 else 500; B = 500
def data(n):
    X = (p["X_mean"][None, :] + p["X_std"][None, :] * rs.standard_normal((n, 33))).astype(np.float32)
    Y = (p["data"][None, :] + 3 * p["sigma"][None, :] * rs.standard_normal((n, 33))).astype(np.float32)
    return X, Y
X, Y = data(n); VX, VY = data(nv)
ytd = util.Y_transform_data(p["sigma"], "cpu")
yinv = util.Y_invtransform_class(t32(p["y_mean"]), t32(p["y_std"]), t32(p["data"]), "cpu")
args = (t32(p["data"]), torch.tensor(p["cov"], dtype=torch.float64), torch.tensor(np.linalg.inv(p["cov"]), dtype=torch.float64), ytd, yinv, "cpu")
lf, vf = util.Loss_fn(*args), util.Val_metric_fn(*args)
loader = predictor_gpu.BatchLoader(util.ArrayDataset(X, Y), B, shuffle=True, drop_last=True)
vloader = predictor_gpu.BatchLoader(util.ArrayDataset(VX, VY), nv, shuffle=False, drop_last=False)
pred = p["pred"]
pred.outdir = tempfile.mkdtemp()
np.save(os.path.join(pred.outdir, "lr.npy"), 1e-4)
pred.optim = "automatic"
for ne in (10, 100):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pred.train(loader, ne, lf, vloader, vf)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("n %d nv %d: epochs %d (%d steps each): %.1f ms per epoch = %.0f us per step equivalent" % (n, nv, ne, n // B, dt / ne * 1e3, dt / ne / (n // B) * 1e6), flush=True)
import collections
T = collections.Counter()
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[key] += time.perf_counter() - t0; return r
    setattr(obj, name, g)
wrap(trainer.TrainEngine, "validate_enqueue", "validate_enqueue")
wrap(trainer.TrainEngine, "validate_finish", "validate_finish")
wrap(trainer.TrainEngine, "step", "step (enqueue)")
wrap(trainer._Checkpoints, "record", "ckpt.record")
wrap(predictor_gpu.BatchLoader, "epoch_batches", "epoch_batches")
t0 = time.perf_counter(); pred.train(loader, 100, lf, vloader, vf); torch.cuda.synchronize(); tot = time.perf_counter() - t0
print("TIMELINE 100 epochs: total %.1f ms/epoch; " % (tot * 10) + "; ".join("%s %.2f" % (k, v * 10) for k, v in T.items()) + " (ms per epoch)")
