"""Write throughput into ONE growing file from several threads: positional writes (os.pwrite) against copies into a shared
mapping (mmap + numpy copy, which releases the GIL).  usage: io_write_probe.py DIR..."""
import os, sys, time, mmap, numpy as np
from concurrent.futures import ThreadPoolExecutor
def run(dirn, mode, nthreads, piece_mb, total_mb=2048, block_mb=108):
    path = os.path.join(dirn, "iotest.bin")
    if os.path.exists(path): os.remove(path)
    fd = os.open(path, os.O_RDWR | os.O_CREAT)
    src = np.random.randint(0, 255, block_mb << 20, dtype=np.uint8)
    pool = ThreadPoolExecutor(nthreads)
    piece = piece_mb << 20
    def pw(addr, view):
        done = 0
        while done < view.nbytes: done += os.pwrite(fd, view[done:], addr + done)
    def mm(addr, view):
        a0 = addr & ~(mmap.ALLOCATIONGRANULARITY - 1); d = addr - a0
        m = mmap.mmap(fd, view.nbytes + d, offset=a0, access=mmap.ACCESS_WRITE)
        dst = np.frombuffer(m, dtype=np.uint8, count=view.nbytes, offset=d)
        dst[:] = view
        del dst; m.close()
    eof = 4096 + 123
    t0 = time.perf_counter()
    for b in range(total_mb // block_mb):
        if mode == "mmap": os.ftruncate(fd, eof + src.nbytes)
        jobs = [pool.submit(pw if mode == "pwrite" else mm, eof + off, src[off:off + piece]) for off in range(0, src.nbytes, piece)]
        for j in jobs: j.result()
        eof += src.nbytes
    dt = time.perf_counter() - t0
    os.close(fd); os.remove(path)
    return (total_mb // block_mb) * block_mb / 1024 / dt
for dirn in sys.argv[1:]:
    for mode, nt, pm in (("pwrite", 1, 32), ("pwrite", 4, 32), ("pwrite", 8, 8), ("mmap", 1, 32), ("mmap", 4, 16), ("mmap", 8, 8), ("mmap", 12, 4)):
        print(dirn, mode, nt, "threads", pm, "MB pieces: %.2f GB/s" % run(dirn, mode, nt, pm), flush=True)
