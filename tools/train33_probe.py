"""Iteration 0 of the README problem (33-D Gaussian, 10000 + 500 Latin-hypercube points, ChtoModelv2(33,33)) trained by
this package with the seeds / learning rate of tests/golden/train33_run.npz (the live reference's run of the same
thing): per-step losses and per-epoch validation metrics side by side."""
import sys, os, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import readme33
from linna_amd import util, nn

nep = int(sys.argv[1]) if len(sys.argv) > 1 else 300
prob = readme33.problem()
ndim, means, cov = prob["ndim"], prob["means"], prob["cov"]
sigma = np.sqrt(np.diag(cov))
tmp = tempfile.mkdtemp() + "/"
tx, vx = readme33.design(10000, ndim), readme33.design(500, ndim)
np.savetxt(tmp + "train_samples_x.txt", tx); np.save(tmp + "train_samples_y.npy", tx.copy())
np.savetxt(tmp + "val_samples_x.txt", vx); np.save(tmp + "val_samples_y.npy", vx.copy())
np.save(tmp + "lr.npy", readme33.LR)
torch.manual_seed(readme33.SEED)
t0 = time.time()
pred = util.train_NN(None, cov, np.linalg.inv(cov), sigma, tmp, [tmp], means, None, False, True, 2, 16.0, True, None, 1,
                     nn.ChtoModelv2, {"num_epochs": nep, "batch_size": 500}, False)
print("train_NN: %d epochs in %.1f s" % (nep, time.time() - t0))
tl, vm = pred.train_history
g = None
gp = os.path.join(ROOT, "tests", "golden", "train33_run.npz")
if os.path.isfile(gp):
    g = np.load(gp)
    print("first 12 step losses  here:", np.array2string(tl[:12], precision=6))
    print("first 12 step losses   ref:", np.array2string(g["train_losses"][:12], precision=6))
for e in (0, 1, 2, 4, 9, 19, 49, 99, 149, 199, 249, 299):
    if e < len(vm):
        print("epoch %3d  val here %.5e %.3e %.3e" % (e + 1, vm[e, 0], vm[e, 1], vm[e, 2]),
              ("  ref %.5e %.3e %.3e" % tuple(g["val_metrics"][e]) if g is not None and e < len(g["val_metrics"]) else ""))
unit = np.random.RandomState(5).standard_normal((4000, ndim))
yinv = util.Y_invtransform_data(sigma, "cpu")
for T in (16.0, 1.0):
    th = means[None, :] + np.sqrt(T) * sigma[None, :] * unit
    m = yinv(pred.predict(torch.as_tensor(th, dtype=torch.float32))).cpu().numpy()
    res = (m - th) / sigma[None, :]
    print("emulator residual at the T=%d posterior: rms %.3f sigma, max |mean| %.3f sigma" % (T, np.sqrt(np.mean(res ** 2)), np.abs(res.mean(0)).max()),
          ("  (ref last: rms %.3f)" % float(g["last_res_rms_T%d" % T]) if g is not None else ""))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "train33_here.npz"), train_losses=tl, val_metrics=vm)
