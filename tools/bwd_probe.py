#!/usr/bin/env python
"""Training step and lnP-gradient timings with the one-launch dX chain on / off (LINNA_BWD_STREAM)."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    import bench_paths as bp
    bp.training("train_v2_33_33_b500", "ChtoModelv2", 33, 33)
    bp.training("train_v2_26_457_b500", "ChtoModelv2", 26, 457)
    for B in (512, 4096):
        p = bp.problem("ChtoModelv2", 33, 33, False)
        z = torch.randn(B, 33, device="cuda"); o = torch.empty(B, device="cuda"); g = torch.empty(B, 33, device="cuda")
        bp.res["grad_v2_33_33_B%d" % B] = {"us_per_step": bp.timeit(lambda: p["lp"].evaluate_with_grad(z, out=o, grad=g), 100) * 1e6}
    print("RESULT " + json.dumps({k: round(v["us_per_step"], 1) for k, v in bp.res.items()}))
else:
    for flag in ("1", "0"):
        env = dict(os.environ, LINNA_BWD_STREAM=flag)
        out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
        print("LINNA_BWD_STREAM=" + flag, line[0] if line else out.stderr[-800:])
