"""ml_sampler on the README problem (33-D Gaussian) with everything a CPU cross-check needs kept per iteration:
the training / validation points, the seed the network was constructed under, the learning rate, the training
history and the residual of the trained emulator at the true (tempered) posterior.  Usage:
    python tools/e2e33_stage.py <nwalkers> <nepoch> [fixed_lr | 0 = range test]"""
import sys, os, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import readme33
from linna_amd import util, nn, main as lmain
from linna_amd.sampler import ChainStore

nw = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nepoch = int(sys.argv[2]) if len(sys.argv) > 2 else 101
fixed_lr = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
niter = int(sys.argv[4]) if len(sys.argv) > 4 else 4          # > 4: the last iteration's settings repeated
ntrain = int(sys.argv[5]) if len(sys.argv) > 5 else 10000
MODEL = getattr(nn, sys.argv[6]) if len(sys.argv) > 6 else nn.ChtoModelv2
prob = readme33.problem()
ndim, means, cov, init, priors = prob["ndim"], prob["means"], prob["cov"], prob["init"], prob["priors"]
sig = np.sqrt(np.diag(cov))
out = tempfile.mkdtemp() + "/"
dump = os.path.join(ROOT, "gpurun_out", "e2e33")
os.makedirs(dump, exist_ok=True)
hist = {}
orig = util.train_NN

def train_spy(*a, **k):
    outdir_in = a[4]
    it = len(a[5]) - 1
    if fixed_lr > 0:
        np.save(os.path.join(outdir_in, "lr.npy"), fixed_lr)
    torch.manual_seed(readme33.SEED + it)
    t0 = time.time()
    pred = orig(*a, **k)
    hist[it] = (pred.train_history, time.time() - t0, float(np.load(os.path.join(outdir_in, "lr.npy"))))
    return pred
lmain.train_NN = train_spy
np.random.seed(0)
t0 = time.perf_counter()
ext = lambda a: (a + [a[-1]] * niter)[:niter]
temps = ext([4.0, 2.0, 1.0, 1.0])
chain, logp = lmain.ml_sampler_core(ext([ntrain] * 4), ext([500] * 4), ext([2, 2, 5, 4]), ext([5, 5, 10, 15]), ext([0.03, 0.03, 0.02, 0.01]),
                                    ext([0.2] * 4), ext([0.15] * 4), out, readme33.theory, priors, means, cov, init, None, nw, "cuda", None, False,
                                    temps, None, False, 1, None, MODEL, {"trainingoption": 1, "num_epochs": nepoch, "batch_size": 500},
                                    "emcee")
print("ml_sampler %.1f s" % (time.perf_counter() - t0), flush=True)
th = np.asarray(chain)
print("returned chain %s: mean bias max %.3f sigma (median %.3f), std ratio min %.3f max %.3f" % (
    th.shape, np.max(np.abs(th.mean(0) - means) / sig), np.median(np.abs(th.mean(0) - means) / sig),
    np.min(th.std(0) / sig), np.max(th.std(0) / sig)), flush=True)
unit = np.random.RandomState(5).standard_normal((4000, ndim))
yinv = util.Y_invtransform_data(sig, "cpu")
dev = np.abs((th - means) / sig).max(1)
print("returned chain: fraction of samples with some parameter > 5 sigma off: %.3f, median of the row maximum %.2f sigma (Gaussian: 2.4)" % ((dev > 5).mean(), np.median(dev)), flush=True)
for k, T in enumerate([t * t for t in temps]):
    d = out + "iter_%d/" % k
    ch = ChainStore.load(d + "chemcee_256")
    c = np.asarray(ch["chain_transformed"]); c = c[len(c) // 2:].reshape(-1, ndim)
    pred, _ = util.retrieve_model(d, ndim, ndim, MODEL)
    (tl, vm), dt, lr = hist[k]
    line = "iter %d: lr %.2e, %d epochs in %.1f s, val first/last %.4f/%.4f; chain %d steps, mean bias max %.3f sigma (median %.3f), std ratio %.2f-%.2f (tempered: x%.0f);" % (
        k, lr, len(vm), dt, vm[0, 0], vm[-1, 0], len(ch["chain"]), np.max(np.abs(c.mean(0) - means) / sig), np.median(np.abs(c.mean(0) - means) / sig),
        np.min(c.std(0) / sig), np.max(c.std(0) / sig), np.sqrt(T))
    for TT in (T, 1.0):
        thp = means[None, :] + np.sqrt(TT) * sig[None, :] * unit
        m = yinv(pred.predict(torch.as_tensor(thp, dtype=torch.float32))).cpu().numpy()
        res = (m - thp) / sig[None, :]
        line += " residual at T=%d posterior rms %.3f max|mean| %.3f;" % (TT, np.sqrt(np.mean(res ** 2)), np.abs(res.mean(0)).max())
    print(line, flush=True)
    if niter == 4:
        np.savez_compressed(os.path.join(dump, "iter_%d.npz" % k), train_x=np.loadtxt(d + "train_samples_x.txt"),
                            val_x=np.loadtxt(d + "val_samples_x.txt"), train_losses=tl, val_metrics=vm, lr=lr, seed=readme33.SEED + k)
