"""Parameter-gradient GEMMs dW[N][K] = sum_b dY[b][n] X[b][k] (both operands k-major) one at a time: where do the 42 us
of the grouped launch of a ChtoModelv2(26,457) step come from?  (M = outputs N, N = inputs K, contraction = batch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_small import time_gemm
shapes = [(1000, 26, 500), (16, 1000, 500), (500, 16, 500), (500, 1000, 500), (32, 500, 500), (250, 32, 500), (250, 500, 500),
          (64, 250, 500), (125, 64, 500), (125, 250, 500), (500, 125, 500), (457, 500, 500), (457, 457, 500)]
tot = 0.0
for M, N, B in shapes:
    for Bk in (B, 512):
        t = min(time_gemm(M, N, Bk, 1, 1, 0x80) for _ in range(3))
        tiles = ((M + 63) // 64) * ((N + 63) // 64)
        print("dW %4d x %4d, batch %d: %6.1f us  (%3d tiles of 64x64, %.2f GFLOP -> %.1f TF)" % (M, N, Bk, t, tiles, 2e-9 * M * N * Bk, 2e-6 * M * N * Bk / t), flush=True)
    tot += t
