"""Where the epochs of the whole-run measurement (bench.py e2e) spend their time: the same ml_sampler_core call, with every
Predictor.train of the run handed a profile dict (trainer._EpochProf).  Prints one JSON object per training run."""
import contextlib
import io
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(nepoch=200, ntrain=10000, nval=500, nwalkers=128):
    import torch
    from linna_amd import main as lmain, nn, trainer
    runs = []
    inner = trainer.run

    def run(*a, **kw):
        a = list(a)
        prof = a[13] if len(a) > 13 and a[13] is not None else {}
        if len(a) > 13:
            a[13] = prof
        else:
            kw["profile"] = prof
        t0 = time.perf_counter()
        out = inner(*a, **kw)
        torch.cuda.synchronize()
        prof["wall_s"] = time.perf_counter() - t0
        runs.append(prof)
        return out
    trainer.run = run
    rs = np.random.RandomState(0)
    ndim = 33
    means = rs.uniform(size=ndim)
    cov = np.diag(0.1 * rs.uniform(0.2, 1.0, size=ndim))
    init = rs.uniform(size=ndim)
    priors = [{"param": "p%d" % i, "dist": "flat", "arg1": -5.0, "arg2": 5.0} for i in range(ndim)]
    tmp = tempfile.mkdtemp(prefix="linna_e2e_probe_")
    try:
        with contextlib.redirect_stdout(io.StringIO()) as log:
            lmain.ml_sampler_core(
                [ntrain] * 4, [nval] * 4, [2, 2, 5, 4], [5, 5, 10, 15], [0.03, 0.03, 0.02, 0.01], [0.2] * 4, [0.15] * 4, tmp + "/",
                lambda x, outdir: x[1], priors, means, cov, init, None, nwalkers, "cuda", None, False, [4.0, 2.0, 1.0, 1.0],
                nnmodel_in=nn.ChtoModelv2, params={"trainingoption": 1, "num_epochs": nepoch, "batch_size": 500}, method="emcee")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for p in runs:
        keep = {k: (round(v, 5) if isinstance(v, float) else v) for k, v in sorted(p.items())}
        print(json.dumps(keep))
    text = log.getvalue()
    for key in ("bad trainning", "learning rate too large", "early stop", "weight decay too small"):
        print(key, text.count(key))


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:]])
