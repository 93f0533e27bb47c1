"""Wall-clock vs device time of evaluate_with_grad on ChtoModelv2(26,457) with a dense covariance (diagnostic)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.argv = sys.argv[:1]
import torch, bench_paths
p = bench_paths.problem("ChtoModelv2", 26, 457, True)
z = torch.randn(4096, 26, device="cuda"); out = torch.empty(4096, device="cuda"); g = torch.empty(4096, 26, device="cuda")
for _ in range(10): p["lp"].evaluate_with_grad(z, out=out, grad=g)
torch.cuda.synchronize()
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(50): p["lp"].evaluate_with_grad(z, out=out, grad=g)
    e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("rep %d: host enqueue %.0f us/eval, wall %.0f us/eval, device (events) %.0f us/eval" % (rep, (t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6, e0.elapsed_time(e1) / 50 * 1e3))
