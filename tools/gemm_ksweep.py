#!/usr/bin/env python
"""K sweep of the fused GEMM: slope = per-K-tile cost, intercept = launch + prologue + epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import time_gemm
import numpy as np

for (M, N) in [(4096, 512), (512, 64)]:
    for cfg in (1, 2, 3):
        for flags, name in ((0, "full"), (0x30, "mfma-only"), (0x40, "nomfma")):
            Ks = [32, 64, 128, 256, 512, 1024, 2048]
            ts = [min(time_gemm(M, N, K, cfg | flags, iters=30) for _ in range(3)) for K in Ks]
            slope = (ts[-1] - ts[-3]) / ((Ks[-1] - Ks[-3]) / 32)
            print("M=%d N=%d cfg%d %-9s " % (M, N, cfg - 1, name) + " ".join("%6.1f" % t for t in ts) + "  | us/tile %.3f" % slope, flush=True)
