import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from linna_amd.sampler import DeviceChain
for nt, nw, nd in ((16000, 128, 33), (3000, 4096, 33), (4000, 128, 33)):
    dc = DeviceChain()
    x = torch.randn(nt, nw, nd, device="cuda").cumsum(0) * 0.01 + torch.randn(nt, nw, nd, device="cuda")
    dc.append(x)
    for _ in range(2): dc.integrated_time()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): tau = dc.integrated_time()
    torch.cuda.synchronize(); print(nt, nw, nd, "ms per call %.1f" % ((time.perf_counter() - t0) / 5 * 1e3), tau[:2])
    # raw FFT cost
    xs = x[:, :, :8].permute(1, 2, 0).to(torch.float64).contiguous()
    n = 1 << (nt - 1).bit_length()
    for dt in (torch.float64, torch.float32):
        y = xs.to(dt)
        for _ in range(2): f = torch.fft.rfft(y, n=2 * n, dim=2); a = torch.fft.irfft(f * f.conj(), n=2 * n, dim=2)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): f = torch.fft.rfft(y, n=2 * n, dim=2); a = torch.fft.irfft(f * f.conj(), n=2 * n, dim=2)
        torch.cuda.synchronize(); print("   fft pair", dt, "%.2f ms for %d series of %d" % ((time.perf_counter() - t0) / 5 * 1e3, nw * 8, 2 * n))
