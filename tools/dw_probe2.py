import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_small import time_gemm
for M, N, B in [(64, 64, 500), (64, 64, 512), (64, 64, 480), (64, 64, 32), (64, 64, 64), (64, 64, 128), (64, 64, 256), (64,64,1024),
                (500, 1000, 500), (1000, 1000, 500), (1000, 2000, 500), (2000, 2000, 500), (1000, 1000, 512), (2000, 2000, 512)]:
    t = min(time_gemm(M, N, B, 1, 1, 0x80) for _ in range(3))
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    print("dW %4d x %4d, batch %4d: %6.1f us  (%4d tiles, %.1f TF)" % (M, N, B, t, tiles, 2e-6 * M * N * B / t), flush=True)
