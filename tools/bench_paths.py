#!/usr/bin/env python
"""Secondary measurements for DESIGN.md (not the driver's bench line): training step
throughput (BASELINE configs 1/3 shapes), batched HMC leapfrog rate (config 5), dense
log-likelihood serving (config 4 shape) and ChtoModelv2 serving."""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import synth
from linna_amd import nn, util, predictor_gpu, trainer, sampler, _lib

t32 = lambda a: torch.as_tensor(np.asarray(a, np.float32))
res = {}


def problem(kind, nin, nout, dense, seed=1, **kw):
    data, cov, priors = synth.gaussian_problem(nin, nout, seed, dense=dense, cond=1e2)
    X_mean, X_std, y_mean, y_std = synth.transform_constants(nin, nout, seed)
    cls = {"ChtoModelv2": nn.ChtoModelv2, "MLP": nn.MLP}[kind]
    torch.manual_seed(1234)
    model = cls(nin, nout, None, **kw)
    sigma = np.sqrt(np.diag(cov))
    pred = predictor_gpu.Predictor(nin, nout, model=model, device="cuda",
                                   X_transform=util.X_transform_class(t32(X_mean), t32(X_std), "cpu", None),
                                   y_transform=util.Y_transform_class(t32(y_mean), t32(y_std), "cpu"))
    lp = util.Log_prob(t32(data), t32(np.linalg.inv(cov)), pred, util.Y_invtransform_data(sigma, "cpu"),
                       util.Transform(priors), 1.0)
    return dict(model=model, pred=pred, lp=lp, data=data, cov=cov, sigma=sigma, y_mean=y_mean, y_std=y_std,
                X_mean=X_mean, X_std=X_std, priors=priors)


def timeit(fn, n, warm=10):
    # warm up for at least 0.4 s of wall time: after the host-side problem set-up the GPU sits at idle clocks and
    # the first ~50 ms of work run 2-3x slow (tools/grad_timing.py), which swamps a 50-iteration timing window
    t_end = time.perf_counter() + 0.4
    k = 0
    while k < warm or time.perf_counter() < t_end:
        fn(); k += 1
        if k % 16 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def serving(name, kind, nin, nout, dense, B=4096, **kw):
    p = problem(kind, nin, nout, dense, **kw)
    z = torch.randn(B, nin, device="cuda"); out = torch.empty(B, device="cuda")
    dt = timeit(lambda: p["lp"].evaluate(z, out=out), 100)
    flop = B * (2.0 * p["model"].macs_per_eval() + (2.0 * nout * nout if dense else 3.0 * nout))
    res[name] = {"us_per_step": dt * 1e6, "evals_per_s": B / dt, "tflops": flop / dt / 1e12}
    z2 = torch.randn(B, nin, device="cuda")
    g = torch.empty(B, nin, device="cuda")
    dt = timeit(lambda: p["lp"].evaluate_with_grad(z2, out=out, grad=g), 50)
    res[name + "_grad"] = {"us_per_step": dt * 1e6, "evals_per_s": B / dt}


def training(name, kind, nin, nout, n=20000, B=500, **kw):
    p = problem(kind, nin, nout, True, **kw)
    rs = np.random.RandomState(3)
    X = (p["X_mean"][None, :] + p["X_std"][None, :] * rs.standard_normal((n, nin))).astype(np.float32)
    Y = (p["data"][None, :] + 3 * p["sigma"][None, :] * rs.standard_normal((n, nout))).astype(np.float32)
    ytd = util.Y_transform_data(p["sigma"], "cpu")
    yinv = util.Y_invtransform_class(t32(p["y_mean"]), t32(p["y_std"]), t32(p["data"]), "cpu")
    lf = util.Loss_fn(t32(p["data"]), torch.tensor(p["cov"], dtype=torch.float64),
                      torch.tensor(np.linalg.inv(p["cov"]), dtype=torch.float64), ytd, yinv, "cpu")
    loader = predictor_gpu.BatchLoader(util.ArrayDataset(X, Y), B, shuffle=True, drop_last=True)
    for graph in (False, True):
        eng = trainer.TrainEngine(p["pred"], loader, lf, None, use_graph=graph)
        opt = predictor_gpu._AdamWState(p["model"], 1e-4)
        if graph:
            eng.prepare_graph(opt)
        perm = torch.stack(loader.epoch_batches()).to(torch.int32).cuda()
        it = [0]

        def step():
            eng.step(opt, perm[it[0] % len(perm)]); it[0] += 1
        dt = timeit(step, 200, warm=20)
        flop = B * (6.0 * p["model"].macs_per_eval() + 3 * 2.0 * nout * nout)
        res[name + ("_graph" if graph else "_direct")] = {"us_per_step": dt * 1e6, "samples_per_s": B / dt,
                                                          "tflops": flop / dt / 1e12, "loss": float(eng.loss_mean.item())}


def hmc(name, kind, nin, nout, B=4096, **kw):
    p = problem(kind, nin, nout, False, **kw)
    x0 = 0.05 * np.random.RandomState(1).standard_normal((B, nin)).astype(np.float32)
    h = sampler.BatchedHMC(p["lp"], x0)
    dt = timeit(lambda: h.step(5, 1e-3), 30, warm=3)
    res[name] = {"us_per_sample": dt * 1e6, "leapfrog_per_s": 5 * B / dt, "chains": B}


if __name__ == "__main__":
    only = sys.argv[1:] 
    runs = [("serve_mlp4x512_33_diag", lambda n: serving(n, "MLP", 33, 33, False)),
            ("serve_v2_33_33_diag", lambda n: serving(n, "ChtoModelv2", 33, 33, False)),
            ("serve_v2_26_457_dense", lambda n: serving(n, "ChtoModelv2", 26, 457, True)),
            ("serve_mlp4x512_40_1000_dense", lambda n: serving(n, "MLP", 40, 1000, True)),
            ("train_v2_33_33_b500", lambda n: training(n, "ChtoModelv2", 33, 33)),
            ("train_v2_26_457_b500", lambda n: training(n, "ChtoModelv2", 26, 457)),
            ("hmc_mlp4x512_33", lambda n: hmc(n, "MLP", 33, 33))]
    for name, fn in runs:
        if not only or name in only:
            fn(name)
    print(json.dumps(res, indent=1))
