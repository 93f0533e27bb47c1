#!/usr/bin/env python
"""Diagnostic: per-phase cycle stamps of the weight-stream MLP kernel (needs a library built with
LINNA_HIPCC_EXTRA=-DSM_STAMPS).  Usage: python tools/stream_stamps.py [B]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
NW = int(os.environ.get("SM_NW", "8"))
nb = (B + 15) // 16
buf = torch.zeros(nb * NW * 16, dtype=torch.int64, device="cuda")
os.environ["LINNA_FUSED_STAMPS"] = "%x" % buf.data_ptr()
import bench
lp, model, consts = bench.build_problem(torch.device("cuda", 0))
z = torch.randn(B, 33, device="cuda"); out = torch.empty(B, device="cuda")
for _ in range(5): lp.evaluate(z, out=out)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(nb, NW, 16).astype(np.float64)
names = ["start", "prologue+barrier", "L0 steps+publish", "L1 steps+publish", "L2 steps+publish", "L3 steps+publish",
         "last layer steps", "reduce+finish"]
d = np.diff(t[:, :, :len(names)], axis=2)
print("phase                 median cycles   (max over waves, median over blocks)")
for i, n in enumerate(names[1:]):
    print("%-22s %10.0f %10.0f" % (n, np.median(d[:, :, i]), np.median(d[:, :, i].max(1))))
tot = t[:, :, len(names) - 1] - t[:, :, 0]
print("total per wave median %.0f cycles; block span median %.0f" % (np.median(tot), np.median(t[:, :, len(names)-1].max(1) - t[:, :, 0].min(1))))
start = t[:, :, 0].min(1); end = t[:, :, len(names) - 1].max(1)
print("first block start -> last block end: %.0f cycles; block start spread %.0f" % (end.max() - start.min(), start.max() - start.min()))
