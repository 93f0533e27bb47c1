#!/usr/bin/env python
"""Dense-covariance log-likelihood: inverse covariance as the last segment of the whole-network kernel
(default) against network launch + row-dot GEMM (LINNA_DENSE_FUSED=0); evaluation and ensemble iterations."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import numpy as np, torch, time
    import bench_paths as bp
    from linna_amd import sampler
    out = {}
    for name, kind, nin, nout, kw in (("v2_26_457", "ChtoModelv2", 26, 457, {}), ("mlp_40_1000", "MLP", 40, 1000, dict(width=512, depth=4)),
                                      ("v2_40_1000", "ChtoModelv2", 40, 1000, {})):
        p = bp.problem(kind, nin, nout, True, **kw)
        for B in (512, 4096):
            z = torch.randn(B, nin, device="cuda"); o = torch.empty(B, device="cuda")
            out["%s B=%d" % (name, B)] = round(bp.timeit(lambda: p["lp"].evaluate(z, out=o), 200) * 1e6, 1)
        ens = sampler.EnsembleSampler(4096, nin, p["lp"], seed=1)
        ens.set_state(0.05 * np.random.RandomState(7).standard_normal((4096, nin)))
        ens.run(300, store=False); torch.cuda.synchronize()
        t0 = time.perf_counter(); ens.run(500, store=False); torch.cuda.synchronize()
        out["%s mcmc steps/s" % name] = round(500 / (time.perf_counter() - t0))
        out["%s fused move" % name] = bool(ens.fused)
    print("RESULT " + json.dumps(out))
else:
    for flag in ("1", "0"):
        env = dict(os.environ, LINNA_DENSE_FUSED=flag)
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        print("LINNA_DENSE_FUSED=" + flag, line[0] if line else r.stderr[-1500:])
