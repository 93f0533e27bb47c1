"""Does replaying the slice sampler's iteration as a hipGraph shorten it?  One iteration (two one-call half steps) of a
128-walker ensemble with a FIXED split, direct launches against the captured graph.  usage: slice_graph_probe.py [nw]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from linna_amd import sampler, _lib
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lp, model, consts = bench.build_problem(torch.device("cuda", 0))
ens = sampler.SliceEnsembleSampler(nw, 33, lp, seed=1)
ens.randomize_split = False
ens.set_state(0.05 * np.random.RandomState(7).standard_normal((nw, 33)))
ens.run(120, store=False)                         # tunes mu, reaches the one-call path
torch.cuda.synchronize()
assert ens._fast_ok and not ens.tune, (ens._fast_ok, ens.tune)
halves = ens._splits()
seed = C.c_uint64((ens.seed + 0x9E3779B97F4A7C15 * (ens.rank + 1)) & 0xFFFFFFFFFFFFFFFF)
def direct(n):
    for _ in range(n):
        ens._step_fast(halves, seed)
direct(50); torch.cuda.synchronize()
t0 = time.perf_counter(); direct(400); torch.cuda.synchronize(); td = (time.perf_counter() - t0) / 400
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    st = _lib.stream()
    ens._step_fast(halves, seed); side.synchronize()
    _lib.call("linna_graph_begin", st)
    ens._step_fast(halves, seed)
    g = C.c_void_p()
    _lib.call("linna_graph_end", st, C.byref(g))
    for _ in range(50):
        _lib.call("linna_graph_launch", g, st)
    side.synchronize()
    t0 = time.perf_counter()
    for _ in range(400):
        _lib.call("linna_graph_launch", g, st)
    side.synchronize()
    tg = (time.perf_counter() - t0) / 400
print("%d walkers: direct launches %.1f us per iteration, hipGraph replay %.1f us" % (nw, td * 1e6, tg * 1e6))
