"""Phase stamps of the grouped parameter-gradient + AdamW launch of a training step (gemm_group_update_kernel), per
workgroup: entry -> first K tiles requested -> K loop done -> epilogue stores drained, with the CU each workgroup ran on.
Needs the diagnostic build:  python linna_amd/_build.py --stamps --source=gemm.hip -DGEMM_STAMPS
usage: gemm_stamps.py [nin nout [B]]   (default 26 457 500: the bench's `training` workload)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("LINNA_LIB_PATH", os.path.join(ROOT, "linna_amd", "liblinna_hip_stamps.so"))
args = sys.argv[1:]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.argv = sys.argv[:1]
import numpy as np, torch, bench_paths
from bench_paths import *
from linna_amd import _lib
nin, nout = (int(args[0]), int(args[1])) if len(args) >= 2 else (26, 457)
B = int(args[2]) if len(args) >= 3 else 500
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
for i in range(200):
    eng.step(opt, perm[i % len(perm)])
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_ulonglong * (8 * 1024))()
rc = lib.linna_debug_gemm_stamps(buf, 8 * 1024)
assert rc == 0, rc
s = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
s = s[s[:, 0] > 0]
s = s[s[:, 3] > 0]                       # workgroups of the update launch (the extra mean workgroup leaves early)
t0 = s[:, 0].min()
hw = s[:, 4] & 0xFFFFFFFF
xcc = s[:, 4] >> 32
cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | (xcc << 8)      # CU_ID, SE_ID, XCC
print("%d workgroups on %d CUs" % (len(s), len(set(cu.tolist()))))
# (the cycle counters of different XCDs have different origins: spans are taken per XCD)
spans = [(s[xcc == x, 3].max() - s[xcc == x, 0].min()) / 1e3 for x in sorted(set(xcc.tolist()))]
print("per-XCD span, first entry -> last drained: " + " ".join("%.1f" % v for v in spans) + " k cycles")
d = lambda a, b: (s[:, b] - s[:, a]) / 1e3
for name, v in (("prologue (first tiles requested)", d(0, 1)), ("K loop", d(1, 2)), ("epilogue (AdamW, streams, drained)", d(2, 3)), ("whole workgroup", d(0, 3))):
    print("  %-36s min %7.1f  median %7.1f  p90 %7.1f  max %7.1f k cycles" % (name, v.min(), np.median(v), np.percentile(v, 90), v.max()))
# inside the ring loop (wave 0): waiting for its own DMA, at the barrier, issuing + MFMAs; the rest of the K loop is the
# partial last K tile through registers
for name, v in (("  ring: wait for own K tile", s[:, 5] / 1e3), ("  ring: barrier", s[:, 6] / 1e3), ("  ring: issue + fragments + MFMA", s[:, 7] / 1e3),
                ("  partial K tile (register path)", d(1, 2) - (s[:, 5] + s[:, 6] + s[:, 7]) / 1e3)):
    print("  %-36s min %7.1f  median %7.1f  p90 %7.1f  max %7.1f k cycles" % (name, v.min(), np.median(v), np.percentile(v, 90), v.max()))
# workgroups alone on their CU against those that shared it
from collections import Counter
cnt = Counter(cu.tolist())
alone = np.array([cnt[c] == 1 for c in cu.tolist()])
for lab, m in (("alone on its CU", alone), ("sharing its CU", ~alone)):
    if m.any():
        print("  %-18s %4d workgroups: K loop median %.1f k, whole %.1f k" % (lab, m.sum(), np.median(d(1, 2)[m]), np.median(d(0, 3)[m])))
