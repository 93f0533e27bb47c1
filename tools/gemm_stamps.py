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
rbuf = (C.c_ulonglong * (2 * 1024))()
lib.linna_debug_gemm_realtime.restype = C.c_int
assert lib.linna_debug_gemm_realtime(rbuf, 2 * 1024) == 0
rt = np.frombuffer(rbuf, dtype=np.uint64).reshape(1024, 2).astype(np.int64)
keep = (s[:, 0] > 0) & (s[:, 3] > 0)     # workgroups of the update launch (the extra mean workgroup leaves early)
s, rt = s[keep], rt[keep]
# chip-wide timeline (s_memrealtime, 100 MHz): when the workgroups entered and left, relative to the first entry
e_us, x_us = (rt[:, 0] - rt[:, 0].min()) / 100.0, (rt[:, 1] - rt[:, 0].min()) / 100.0
print("realtime: entry  median %.2f p90 %.2f max %.2f us;  end  median %.2f p90 %.2f max %.2f us  (launch = first entry -> last end: %.2f us)" % (
    np.median(e_us), np.percentile(e_us, 90), e_us.max(), np.median(x_us), np.percentile(x_us, 90), x_us.max(), x_us.max()))
late = e_us > 1.0
print("  %d workgroups entered more than 1 us after the first (median entry of those %.2f us)" % (late.sum(), np.median(e_us[late]) if late.any() else 0.0))
t0 = s[:, 0].min()
hw = s[:, 4] & 0xFFFFFFFF
xcc = s[:, 4] >> 32
cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | (xcc << 8)      # CU_ID, SE_ID, XCC
print("%d workgroups on %d CUs" % (len(s), len(set(cu.tolist()))))
# (the cycle counters of different XCDs have different origins: spans are taken per XCD)
spans = [(s[xcc == x, 3].max() - s[xcc == x, 0].min()) / 1e3 for x in sorted(set(xcc.tolist()))]
print("per-XCD span, first entry -> last drained: " + " ".join("%.1f" % v for v in spans) + " k cycles")
# when the workgroups entered, relative to the first entry of their XCD (dispatch ramp; late entries = second round on a busy CU)
ent = np.concatenate([(s[xcc == x, 0] - s[xcc == x, 0].min()) / 1e3 for x in sorted(set(xcc.tolist()))])
end = np.concatenate([(s[xcc == x, 3] - s[xcc == x, 0].min()) / 1e3 for x in sorted(set(xcc.tolist()))])
print("entry after the XCD's first entry: median %.1f  p90 %.1f  max %.1f k cycles;  drained: median %.1f  p90 %.1f  max %.1f k cycles" % (
    np.median(ent), np.percentile(ent, 90), ent.max(), np.median(end), np.percentile(end, 90), end.max()))
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
