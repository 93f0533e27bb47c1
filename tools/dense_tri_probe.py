"""The dense log-likelihood segment over the lower-triangular factor under the three linna_dense_tri modes (0 full factor,
1 second column pass from row 512, 2 balanced 64-column blocks): the products skipped are zeros, so lnP must be
BIT-identical between the modes, and the launch shorter.  One process (the mode applies to log-probability objects created
after it is set).  usage: dense_tri_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch, bench
from linna_amd import _lib
import cases
from test_gpu_serving import build_logprob
z = np.random.RandomState(3).standard_normal((777, 40)).astype(np.float32) * 0.5
got = {}
for mode in (0, 1, 2):
    _lib.load().linna_dense_tri(mode)
    r = bench.secondary_serving(torch.device("cuda", 0), "ChtoModelv2", 40, 1000, True)
    got[mode] = build_logprob("v2_40_1000")[0](z, returntorch=False)
    print("linna_dense_tri(%d): %.2f us per launch; %.1f %% of the fp32-MFMA peak on the FLOP executed, %.1f %% priced at 2 nout^2" % (
        mode, r["us_per_launch"], 100 * r["frac"], 100 * r["frac_priced"]), flush=True)
print("bit-identical lnP over 777 walkers: mode 1 vs 0:", bool(np.array_equal(got[1], got[0])), " mode 2 vs 0:", bool(np.array_equal(got[2], got[0])),
      " max |lnP|", float(np.abs(got[0]).max()))
