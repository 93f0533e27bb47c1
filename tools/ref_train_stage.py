"""Container-only cross-check: the LIVE reference's train_NN on the training points an e2e33_stage.py run dumped
(gpurun_out/e2e33/iter_k.npz), same seed / learning rate / epochs -> its validation trajectory next to the HIP one."""
import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import _ref_import, readme33
rnn, rutil, rpred, rhmc = _ref_import.import_reference()
import torch
torch.set_num_threads(6)
it = int(sys.argv[1]); nep = int(sys.argv[2]) if len(sys.argv) > 2 else 101
prob = readme33.problem()
means, cov = prob["means"], prob["cov"]
sigma = np.sqrt(np.diag(cov))
base = tempfile.mkdtemp() + "/"
dirs = []
for k in range(it + 1):
    g = np.load(os.path.join(ROOT, "gpurun_out", "e2e33", "iter_%d.npz" % k))
    d = base + "iter_%d/" % k
    os.makedirs(d)
    tx, vx = g["train_x"], g["val_x"]
    if k < it:      # earlier iterations contributed only their own 10000 / 500 points
        pass
    # the dump holds the CONCATENATED set of iterations 0..k: keep the last 10000 / 500 as this iteration's own
    own_t, own_v = tx[-10000:], vx[-500:]
    np.savetxt(d + "train_samples_x.txt", own_t); np.save(d + "train_samples_y.npy", own_t.copy())
    np.savetxt(d + "val_samples_x.txt", own_v); np.save(d + "val_samples_y.npy", own_v.copy())
    dirs.append(d)
g = np.load(os.path.join(ROOT, "gpurun_out", "e2e33", "iter_%d.npz" % it))
np.save(dirs[-1] + "lr.npy", float(g["lr"]))
cap = {}
orig = rpred.Predictor.train
def spy(self, *a, **k):
    cap["ret"] = orig(self, *a, **k); return cap["ret"]
rpred.Predictor.train = spy
class _S(object): pass
T = [16.0, 4.0, 1.0, 1.0][it]
torch.manual_seed(int(g["seed"]))
rutil.train_NN(_S(), cov, np.linalg.inv(cov), sigma, dirs[-1], dirs, means, None, False, True, 2, T, False, None, 1,
               rnn.ChtoModelv2, {"num_epochs": nep, "batch_size": 500}, False)
tl, vm = cap["ret"]
print("\nfirst 8 step losses ref :", np.array2string(np.asarray(tl[:8]), precision=6))
print("first 8 step losses here:", np.array2string(g["train_losses"][:8], precision=6))
for e in (0, 1, 2, 4, 9, 19, 29, 39, 49, 59, 79, 100):
    if e < len(vm):
        print("epoch %3d  val ref %.5e   here %.5e" % (e + 1, vm[e][0], g["val_metrics"][e, 0]))
