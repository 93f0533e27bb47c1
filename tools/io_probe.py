import torch, time, os, numpy as np
x = torch.randn(100, 4096, 33, device="cuda")
torch.cuda.synchronize()
for pin in (False, True):
    ts = []
    for i in range(6):
        t0 = time.perf_counter()
        if pin:
            h = torch.empty(x.shape, dtype=x.dtype, pin_memory=True)
            t1 = time.perf_counter()
            h.copy_(x, non_blocking=True); torch.cuda.synchronize()
        else:
            t1 = t0
            h = x.to("cpu")
        ts.append((time.perf_counter() - t0, t1 - t0))
    print("D2H 54 MB pinned=%s: %s ms (alloc %s ms)" % (pin, [round(1e3 * a, 1) for a, b in ts], [round(1e3 * b, 1) for a, b in ts]))
h = x.to("cpu").numpy()
mv = memoryview(h.reshape(-1)).cast("B")
for d in ("/dev/shm", "/tmp"):
    fd = os.open(d + "/io_probe.bin", os.O_CREAT | os.O_RDWR)
    ts = []
    for i in range(8):
        t0 = time.perf_counter(); os.pwrite(fd, mv, i * mv.nbytes); ts.append(time.perf_counter() - t0)
    os.close(fd); os.remove(d + "/io_probe.bin")
    print("pwrite 54 MB to %s: %s ms" % (d, [round(1e3 * t, 1) for t in ts]))
t0 = time.perf_counter(); c = h.copy(); print("host memcpy 54 MB: %.1f ms" % (1e3 * (time.perf_counter() - t0)))
