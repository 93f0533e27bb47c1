#!/usr/bin/env python
"""Diagnostic: per-phase cycle stamps of the fused MLP kernel (s_memtime, 100 MHz-independent
shader clock).  Usage: python tools/fused_stamps.py [B]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nb = (B + 15) // 16
buf = torch.zeros(nb * 8 * 16, dtype=torch.int64, device="cuda")
os.environ["LINNA_FUSED_STAMPS"] = "%x" % buf.data_ptr()
import bench
lp, model, consts = bench.build_problem(torch.device("cuda", 0))
z = torch.randn(B, 33, device="cuda"); out = torch.empty(B, device="cuda")
for _ in range(5): lp.evaluate(z, out=out)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(nb, 8, 16).astype(np.float64)
names = ["start", "bias+prologue", "ring fill issued"] + sum([["L%d mfma" % l, "L%d store+bar" % l] for l in range(4)], []) + ["last mfma", "finish"]
d = np.diff(t[:, :, :len(names)], axis=2)
print("phase                 median cycles   (max over waves, median over blocks)")
for i, n in enumerate(names[1:]):
    print("%-22s %10.0f %10.0f" % (n, np.median(d[:, :, i]), np.median(d[:, :, i].max(1))))
tot = t[:, :, len(names) - 1] - t[:, :, 0]
print("total per wave median %.0f cycles; block span median %.0f" % (np.median(tot), np.median(t[:, :, len(names)-1].max(1) - t[:, :, 0].min(1))))
