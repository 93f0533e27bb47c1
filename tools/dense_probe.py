"""Profile/timing target: evaluations of the config-4 shape (MLP 4x512 (40,1000), dense 1000x1000 inverse covariance)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.argv = sys.argv[:1]
import torch, bench_paths
p = bench_paths.problem("MLP", 40, 1000, True)
z = torch.randn(4096, 40, device="cuda"); out = torch.empty(4096, device="cuda")
for _ in range(30): p["lp"].evaluate(z, out=out)
torch.cuda.synchronize()
if os.environ.get("DENSE_TIMING"):
    for rep in range(6):
        t0 = time.perf_counter()
        for _ in range(100): p["lp"].evaluate(z, out=out)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("rep %d: host enqueue %.1f us/eval, total %.1f us/eval" % (rep, (t1 - t0) * 1e4, (t2 - t0) * 1e4))
