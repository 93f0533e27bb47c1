"""Profile target: ONE serving workload, one launch per step, nothing else (so that a rocprofv3 kernel-stats row isolates
it).  Usage: serve_probe.py <kind> <nin> <nout> <dense 0|1> [nwalkers [iters]]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = sys.argv[1:]
sys.argv = sys.argv[:1]
import torch, bench
kind, nin, nout, dense = args[0], int(args[1]), int(args[2]), bool(int(args[3]))
nw = int(args[4]) if len(args) > 4 else 4096
iters = int(args[5]) if len(args) > 5 else 2000
torch.cuda.set_device(0)
r = bench.secondary_serving(torch.device("cuda", 0), kind, nin, nout, dense, nw, iters)
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()})
