"""The dense log-likelihood's second pass over the lower-triangular factor starts at k = 512 (net_stream.hip: NsSeg::zext for a
WIDE segment): the products it skips are zeros, so lnP must be BIT-identical to the full pass (LINNA_DENSE_TRI=0), and the
launch shorter.  Runs itself twice as child processes (the switch is read once per process).  usage: dense_tri_probe.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
    import numpy as np, torch, bench
    sys.argv = sys.argv[:1]
    r = bench.secondary_serving(torch.device("cuda", 0), "ChtoModelv2", 40, 1000, True)
    import cases
    from test_gpu_serving import build_logprob
    lp = build_logprob("v2_40_1000")[0]
    z = np.random.RandomState(3).standard_normal((777, 40)).astype(np.float32) * 0.5
    got = lp(z, returntorch=False)
    np.save(sys.argv[0] + ".%s.npy" % os.environ.get("LINNA_DENSE_TRI", "1"), got)
    print("LINNA_DENSE_TRI=%s: %.2f us per launch (%.1f %% of peak)" % (os.environ.get("LINNA_DENSE_TRI", "1"), r["us_per_launch"], 100 * r["frac"]))
else:
    import numpy as np
    for v in ("0", "1"):
        e = dict(os.environ, LINNA_DENSE_TRI=v)
        print(subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e, capture_output=True, text=True).stdout.strip().splitlines()[-1])
    a, b = (np.load(os.path.abspath(__file__) + ".%s.npy" % v) for v in ("0", "1"))
    print("bit-identical lnP over 777 walkers:", bool(np.array_equal(a, b)), " max |lnP|", float(np.abs(a).max()))
    for v in ("0", "1"):
        os.remove(os.path.abspath(__file__) + ".%s.npy" % v)
