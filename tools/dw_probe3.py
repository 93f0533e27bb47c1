"""dW GEMMs (k-major x k-major, K = batch 500): 64x64 tiles (cfg 2 -> flag 2) against 128x64 tiles (cfg 1 -> flag 1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_small import time_gemm
for M, N in [(64, 64), (128, 64), (500, 1000), (457, 500), (457, 457), (250, 500), (1000, 26), (1000, 1000)]:
    for flag, name in ((0x80 | 2, "64x64"), (0x80 | 1, "128x64")):
        t = min(time_gemm(M, N, 500, 1, 1, flag) for _ in range(3))
        print("dW %4d x %4d batch 500, %-7s tiles: %6.1f us  (%.1f TF)" % (M, N, name, t, 2e-6 * M * N * 500 / t), flush=True)
