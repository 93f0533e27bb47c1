#!/usr/bin/env python
"""Diagnostic: per-segment cycle stamps of net_stream_kernel (the diagnostic build: python linna_amd/_build.py --stamps).  Usage: python tools/ns_stamps.py [mlp|v2] [B]   (NS_STAMPS_GRAD=1: lnP + gradient in one launch)
Stamp order: start, after prologue barrier, after every segment's last barrier, after the loop, end."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("LINNA_LIB_PATH", os.path.join(ROOT, "linna_amd", "liblinna_hip_stamps.so"))   # python linna_amd/_build.py --stamps
which = sys.argv[1] if len(sys.argv) > 1 else "v2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
nb = (B + 3) // 4                       # room for the smallest engine's grid (4 rows per workgroup); unused blocks stay zero
buf = torch.zeros(nb * 8 * 32, dtype=torch.int64, device="cuda")
os.environ["LINNA_FUSED_STAMPS"] = "%x" % buf.data_ptr()
if which == "mlp":
    import bench
    lp, model, consts = bench.build_problem(torch.device("cuda", 0))
    nin = 33
else:
    sys.argv = sys.argv[:1]
    import bench_paths
    nin, nout = (33, 33) if which == "v2" else (26, 457)
    p = bench_paths.problem("ChtoModelv2", nin, nout, which != "v2")
    lp = p["lp"]
z = torch.randn(B, nin, device="cuda"); out = torch.empty(B, device="cuda")
if os.environ.get("NS_STAMPS_STRETCH"):              # the one-launch stretch half step of an ensemble of 2 B walkers instead
    from linna_amd import sampler
    ens = sampler.EnsembleSampler(2 * B, nin, lp, seed=1)
    ens.set_state(0.05 * np.random.RandomState(7).standard_normal((2 * B, nin)))
    ens.run(6, store=False)
elif os.environ.get("NS_STAMPS_GRAD"):               # the one-launch gradient (HMC) instead of the evaluation
    g = torch.empty(B, nin, device="cuda")
    for _ in range(5): lp.evaluate_with_grad(z, out=out, grad=g)
else:
    for _ in range(5): lp.evaluate(z, out=out)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(nb, 8, 32).astype(np.float64)
t = t[t[:, 0, 0] > 0]                      # the workgroups that ran
print("%d workgroups" % len(t))
n = int((t[0, 0] > 0).sum())
d = np.diff(t[:, :, :n], axis=2)
print("phase   median cycles   (max over waves, median over blocks)")
for i in range(n - 1):
    print("%2d -> %2d %10.0f %10.0f" % (i, i + 1, np.median(d[:, :, i]), np.median(d[:, :, i].max(1))))
print("total per wave median %.0f cycles" % np.median(t[:, :, n - 1] - t[:, :, 0]))
if os.environ.get("NS_STAMPS_WAVES"):
    # per wave: when it reaches every stamp, relative to the workgroup's first wave there (median over blocks) -- who lags in a run
    rel = t[:, :, :n] - t[:, :, :n].min(axis=1, keepdims=True)
    print("lag behind the first wave at each stamp (cycles, median over blocks), one row per wave:")
    for w in range(8):
        print("wave %d " % w + " ".join("%6.0f" % np.median(rel[:, w, i]) for i in range(n)))
    longest = int(np.argmax(np.median(d, axis=(0, 1))))
    print("duration of phase %d -> %d per wave (median over blocks): %s" % (longest, longest + 1, " ".join("%.0f" % np.median(d[:, w, longest]) for w in range(8))))
