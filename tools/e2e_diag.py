"""After an ml_sampler run on the README problem: per iteration, the emulator's residual at its own
posterior samples (in units of the data sigma) and the bias of the chain mean."""
import sys, os, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from linna_amd.main import ml_sampler
from linna_amd import util, nn
from linna_amd.sampler import ChainStore
np.random.seed(0)
ndim = 33
means = np.random.uniform(size=ndim)
cov = np.diag(0.1 * np.random.uniform(size=ndim))
sig = np.sqrt(np.diag(cov))
init = np.random.uniform(size=ndim)
priors = [{"param": "p%d" % i, "dist": "flat", "arg1": -5.0, "arg2": 5.0} for i in range(ndim)]
def theory(x, outdir):
    return x[1]
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nepoch = int(sys.argv[2]) if len(sys.argv) > 2 else 400
out = tempfile.mkdtemp() + "/"
t0 = time.perf_counter()
chain, logp = ml_sampler(out, theory, priors, means, cov, init, None, nw, gpunode=None, nepoch=nepoch, method="emcee")
print("ml_sampler %.1f s" % (time.perf_counter() - t0), flush=True)
th = np.asarray(chain)
print("returned chain %s: mean bias max %.3f sigma (median %.3f), std ratio min %.3f max %.3f" % (
    th.shape, np.max(np.abs(th.mean(0) - means) / sig), np.median(np.abs(th.mean(0) - means) / sig),
    np.min(th.std(0) / sig), np.max(th.std(0) / sig)), flush=True)
for k in range(4):
    d = out + "iter_%d/" % k
    ch = ChainStore.load(d + "chemcee_256")
    th = np.asarray(ch["chain_transformed"]); th = th[len(th) // 2:].reshape(-1, ndim)
    sub = th[np.random.RandomState(1).randint(0, len(th), 4000)]
    pred, yinv = util.retrieve_model(d, ndim, ndim, nn.ChtoModelv2)
    m = yinv(pred.predict(torch.as_tensor(sub, dtype=torch.float32))).cpu().numpy()
    res = (m - sub) / sig
    tx = np.loadtxt(d + "train_samples_x.txt") if os.path.isfile(d + "train_samples_x.txt") else None
    print("iter %d: train pts %s  chain mean bias max %.3f sigma (median %.3f);  emulator residual at posterior: rms %.3f sigma, max |mean residual| %.3f sigma"
          % (k, None if tx is None else tx.shape, np.max(np.abs(th.mean(0) - means) / sig), np.median(np.abs(th.mean(0) - means) / sig),
             np.sqrt(np.mean(res ** 2)), np.max(np.abs(res.mean(0)))), flush=True)
