#!/usr/bin/env python
"""Randomised shapes through the whole-network kernel (all engines, diagonal and dense covariance, evaluation,
gradient, training forward + one-launch dX chain) against the numpy oracle / the GEMM-chain paths.
usage: fuzz_net_stream.py [nconfigs] [seed0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np, torch
import cases
from test_gpu_serving import build_logprob, _custom_problem
from oracle import likelihood
from linna_amd import nn, _lib

def run(ncfg, seed0):
    """Returns the number of failing configurations."""
    bad = 0
    for it in range(ncfg):
        rs = np.random.RandomState(seed0 + it)
        nin = int(rs.choice([1, 2, 7, 16, 33, 40, 64, 65, 130, 200]))
        nout = int(rs.choice([1, 3, 16, 33, 64, 65, 100, 129, 250, 257, 457, 700]))
        width = int(rs.choice([8, 16, 24, 48, 64, 100, 128, 200, 256, 300, 512, 513, 700, 1000, 1024]))
        depth = int(rs.randint(1, 5))
        dense = bool(rs.randint(0, 2))
        B = int(rs.choice([1, 3, 4, 5, 16, 17, 63, 300, 600]))
        rows = rs.choice(["", "4", "8", "16"])
        kind = str(rs.choice(["MLP", "MLP", "ChtoModelv2", "ChtoModelsimple"]))
        if kind != "MLP":
            nout = int(rs.choice([1, 2, 3, 5, 8, 10, 16, 25, 30, 31, 33, 64, 100, 457]))
        tag = "cfg %d: %s nin %d nout %d width %d depth %d dense %d B %d rows %s" % (seed0 + it, kind, nin, nout, width, depth, dense, B, rows or "auto")
        if rows:
            _lib.engine_rows(int(rows))
        else:
            _lib.engine_rows(0)
        try:
            prob = _custom_problem(nin, nout, 7000 + seed0 + it, width, depth, dense=dense)
            if kind != "MLP":
                import synth
                prob["kind"], prob["kw"] = kind, {}
                prob["weights"] = synth.weights(kind, nin, nout, 7000 + seed0 + it)
            if nin > 2 and rs.randint(0, 3) == 0:                 # log10 of some positive-range inputs (util.py:483-497)
                cols = sorted(set(int(c) for c in rs.choice(nin, size=min(3, nin), replace=False)))
                prob["dolog10"] = cols
                for c in cols:
                    prob["priors"][c] = {"param": "p%d" % c, "dist": "flat", "arg1": 0.1, "arg2": 2.0}
                tag += " log10 %s" % cols
            lp = build_logprob(None, 2.0, prob=prob)[0]
            emu = cases.oracle_emulator(prob)
            z = (0.6 * rs.standard_normal((B, nin))).astype(np.float32)
            zd = torch.as_tensor(z, device="cuda")
            ref = likelihood.log_prob(z, emu, prob["priors"], prob["data"], prob["invcov"], 2.0, dtype=np.float64)
            got = lp.evaluate(zd).cpu().numpy()
            e1 = np.max(np.abs(got - ref) / (np.abs(ref) + 1e-3))
            lnp, g = lp.evaluate_with_grad(zd)
            _, gref = likelihood.grad_log_prob(z.astype(np.float64), emu, prob["priors"], prob["data"], prob["invcov"], 2.0, dtype=np.float64)
            e2 = np.max(np.abs(lnp.cpu().numpy() - ref) / (np.abs(ref) + 1e-3))
            e3 = np.max(np.abs(g.cpu().numpy()[:, :nin] - gref)) / (np.abs(gref).max() + 1e-9)
            # training forward / backward: fused dX chain against the GEMM chain
            torch.manual_seed(it)
            cls = {"MLP": nn.MLP, "ChtoModelv2": nn.ChtoModelv2, "ChtoModelsimple": nn.ChtoModelsimple}[kind]
            ma = cls(nin, nout, None, **prob["kw"]); ma.load_state_dict(prob["weights"]); ma.cuda()
            mb = cls(nin, nout, None, **prob["kw"]); mb.load_state_dict(prob["weights"]); mb.cuda()
            x = torch.randn(B, nin, device="cuda"); dout = torch.randn(B, nout, device="cuda") / B
            os.environ["LINNA_BWD_STREAM"] = "0"
            mb.forward(x); dxb = mb.backward(dout, param_grads=True, need_dx=True)[:, :nin].clone()
            os.environ.pop("LINNA_BWD_STREAM")
            ya = ma.forward(x); dxa = ma.backward(dout, param_grads=True, need_dx=True)[:, :nin]
            e4 = float((dxa - dxb).abs().max() / (dxb.abs().max() + 1e-12))
            e5 = max(float((ma.grad_dict()[k] - mb.grad_dict()[k]).abs().max() / (mb.grad_dict()[k].abs().max() + 1e-12)) for k in ma.grad_dict())
            # gradient: a row whose path crosses a ReLU kink within float32 rounding of the float64 oracle differs by a
            # finite amount while lnP agrees; at most one such row (or 0.5 % of the rows) is not counted as a failure
            rowerr_all = np.abs(g.cpu().numpy()[:, :nin] - gref).max(1) / (np.abs(gref).max() + 1e-9)
            kinks = int((rowerr_all > 5e-3).sum())
            # ... and a row whose forward path passes within 1e-6 of a ReLU kink is not counted at all: the float64 and the
            # float32 forward of the oracle disagree on which units are zero there, or a positive unit is that small
            import parity                        # (tests/parity.py: THE ReLU-kink criterion, one formula for tests and tools)
            near_kink = lambda r: parity.near_relu_kink(z[r], emu, prob["priors"])
            explained = sum(1 for r in np.where(rowerr_all > 5e-3)[0] if near_kink(int(r)))
            ok = e1 < 1e-3 and e2 < 1e-3 and kinks - explained <= max(1, B // 200) and e4 < 2e-3 and e5 < 2e-3
            if kinks:
                rowerr = np.abs(g.cpu().numpy()[:, :nin] - gref).max(1) / (np.abs(gref).max() + 1e-9)
                worst = int(np.argmax(rowerr))
                # the same row through the float32 oracle and with z nudged: a ReLU kink shows as an unstable reference
                _, g32 = likelihood.grad_log_prob(z[worst:worst + 1], emu, prob["priors"], prob["data"], prob["invcov"], 2.0, dtype=np.float32)
                zn = z[worst:worst + 1].astype(np.float64) * (1 + 1e-6)
                _, gn = likelihood.grad_log_prob(zn, emu, prob["priors"], prob["data"], prob["invcov"], 2.0, dtype=np.float64)
                print("     rows > 5e-3: %d of %d (%d on a ReLU kink); worst row %d: |gpu-ref64| %.2e |ref32-ref64| %.2e |ref64(z(1+1e-6))-ref64| %.2e (of max|g|)" % (
                    int((rowerr > 5e-3).sum()), B, explained, worst, rowerr[worst], np.abs(g32[0] - gref[worst]).max() / np.abs(gref).max(),
                    np.abs(gn[0] - gref[worst]).max() / np.abs(gref).max()), flush=True)
            print(("ok   " if ok else "BAD  ") + tag + "  eval %.1e grad-lnP %.1e grad %.1e dX %.1e dW %.1e  states %s" % (e1, e2, e3, e4, e5, ma.stream_state()), flush=True)
            bad += 0 if ok else 1
        except Exception as e:
            print("EXC  " + tag + "  " + repr(e)[:300], flush=True)
            bad += 1
    for k in ("LINNA_NS_ROWS", "LINNA_BWD_STREAM"):
        os.environ.pop(k, None)
    print("fuzz: %d configurations, %d bad" % (ncfg, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
