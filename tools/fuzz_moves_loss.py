#!/usr/bin/env python
"""Randomised checks of (1) the one-launch stretch half step against propose / evaluate / accept (bit for bit) over
network shapes, ensemble sizes (engines), diagonal and dense covariance; (2) the one-launch chi2-ratio loss against
the five-launch path over nout <= 64, batch sizes, masked sentinels.  usage: fuzz_moves_loss.py [n] [seed0]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np, torch
from test_gpu_serving import build_logprob, _custom_problem
from linna_amd import sampler, _lib


def moves(n, seed0):
    bad = 0
    for it in range(n):
        rs = np.random.RandomState(seed0 + it)
        nin = int(rs.choice([2, 5, 16, 33, 64])); nout = int(rs.choice([1, 8, 33, 64, 100, 300, 600]))
        width = int(rs.choice([16, 64, 128, 256, 512, 1000])); depth = int(rs.randint(1, 4)); dense = bool(rs.randint(0, 2))
        nw = int(rs.choice([4, 6, 34, 128, 600, 2100, 4096]))
        tag = "move cfg %d: nin %d nout %d width %d depth %d dense %d nw %d" % (seed0 + it, nin, nout, width, depth, dense, nw)
        try:
            prob = _custom_problem(nin, nout, 11000 + seed0 + it, width, depth, dense=dense)
            lp = build_logprob(None, 2.0, prob=prob)[0]
            x0 = (0.3 * rs.standard_normal((nw, nin))).astype(np.float32)
            a = sampler.EnsembleSampler(nw, nin, lp, seed=21 + it)
            b = sampler.EnsembleSampler(nw, nin, lp, seed=21 + it, fused=False)
            a.set_state(x0); b.set_state(x0)
            for _ in range(4):
                a.step(); b.step()
            torch.cuda.synchronize()
            ok = a.fused is True and torch.equal(a.coords, b.coords) and torch.equal(a.logp, b.logp) and torch.equal(a.naccept, b.naccept)
            ok = ok and bool(torch.isfinite(a.logp).all()) and 0 < int(a.naccept.sum())
            print(("ok   " if ok else "BAD  ") + tag + "  accepted %d" % int(a.naccept.sum()), flush=True)
            bad += 0 if ok else 1
        except Exception as e:
            print("EXC  " + tag + "  " + repr(e)[:300], flush=True); bad += 1
    return bad


def slices(n, seed0):
    """Ensemble slice sampler: trial points built in the evaluation kernel's prologue against points written
    to memory first (linna_slice_points + gated evaluation): same states, bit for bit."""
    bad = 0
    for it in range(n):
        rs = np.random.RandomState(seed0 + it)
        nin = int(rs.choice([2, 5, 16, 33, 64])); nout = int(rs.choice([1, 8, 33, 100, 300]))
        width = int(rs.choice([16, 64, 256, 512])); depth = int(rs.randint(1, 4)); dense = bool(rs.randint(0, 2))
        nw = int(rs.choice([4, 10, 34, 128, 600, 2100]))
        tag = "slice cfg %d: nin %d nout %d width %d depth %d dense %d nw %d" % (seed0 + it, nin, nout, width, depth, dense, nw)
        try:
            prob = _custom_problem(nin, nout, 12000 + seed0 + it, width, depth, dense=dense)
            lp = build_logprob(None, 2.0, prob=prob)[0]
            x0 = (0.3 * rs.standard_normal((nw, nin))).astype(np.float32)
            a = sampler.SliceEnsembleSampler(nw, nin, lp, seed=5 + it)
            b = sampler.SliceEnsembleSampler(nw, nin, lp, seed=5 + it)
            b.fused_points = False
            a.set_state(x0); b.set_state(x0)
            for _ in range(2):
                a.step(); b.step()
            torch.cuda.synchronize()
            ok = a.fused_points is True and torch.equal(a.coords, b.coords) and torch.equal(a.logp, b.logp) and a.mu == b.mu
            ok = ok and bool(torch.isfinite(a.logp).all())
            print(("ok   " if ok else "BAD  ") + tag + "  mu %.3f evals/walker %.1f" % (a.mu, a.neval / (2.0 * nw)), flush=True)
            bad += 0 if ok else 1
        except Exception as e:
            print("EXC  " + tag + "  " + repr(e)[:300], flush=True); bad += 1
    return bad


def loss(n, seed0):
    bad = 0
    lib = _lib.load()
    for it in range(n):
        rs = np.random.RandomState(seed0 + it)
        nout = int(rs.choice([1, 2, 3, 7, 16, 33, 40, 63, 64])); B = int(rs.choice([1, 3, 4, 5, 64, 500, 777])); ntot = B + int(rs.randint(0, 50))
        ld = _lib.ld4(nout)
        f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32), device="cuda")
        A = rs.standard_normal((nout, nout)); cinv = A @ A.T / nout + np.eye(nout)
        keep = dict(sigma=f32(rs.uniform(0.5, 2, nout)), ymean=f32(rs.standard_normal(nout)), ystd=f32(rs.uniform(0.5, 2, nout)),
                    dn=f32(rs.standard_normal(nout)), cinv=f32(np.pad(cinv, ((0, 0), (0, ld - nout)))))
        Y = rs.standard_normal((ntot, nout)).astype(np.float32)
        if ntot > 2: Y[1, 0] = 1e10; Y[2, nout - 1] = 1e-30
        d = _lib.LossDesc(); d.nout = nout
        d.sigma, d.ymean, d.ystd, d.data_norm = (_lib.ptr(keep[k]) for k in ("sigma", "ymean", "ystd", "dn"))
        d.Cinv, d.ldc = _lib.ptr(keep["cinv"]), ld
        Yd, pred = f32(Y), f32(np.pad(rs.standard_normal((B, nout)), ((0, 0), (0, ld - nout))))
        rows = torch.as_tensor(rs.permutation(ntot)[:B].astype(np.int32), device="cuda")
        den = f32(rs.uniform(1.0, 3.0, ntot))
        res = []
        for fused in ("1", "0"):
            os.environ["LINNA_LOSS_FUSED"] = fused
            ctx = C.c_void_p(); _lib.check(lib.linna_ctx_create(0, C.byref(ctx)))         # the switch is read per context
            scratch = torch.zeros(lib.linna_loss_scratch_bytes(B, nout) // 4 + 4, device="cuda")
            lr, lm, dp = torch.zeros(B, device="cuda"), torch.zeros(1, device="cuda"), torch.full((B, ld), 7.0, device="cuda")
            for _ in range(2):      # twice: the arrival counter must have reset itself
                _lib.call("linna_chi2_ratio_loss_fwd_bwd", ctx, C.byref(d), _lib.ptr(pred), ld, _lib.ptr(Yd), Yd.stride(0), _lib.ptr(den),
                          _lib.iptr(rows), B, _lib.ptr(scratch), _lib.ptr(lr), _lib.ptr(lm), _lib.ptr(dp), ld, 1.0 / B, _lib.stream())
            torch.cuda.synchronize()
            res.append((lr.cpu().numpy(), float(lm), dp.cpu().numpy()))
            _lib.check(lib.linna_ctx_destroy(ctx))
        os.environ.pop("LINNA_LOSS_FUSED")
        (a_lr, a_lm, a_dp), (b_lr, b_lm, b_dp) = res
        e = [np.abs(a_lr - b_lr).max() / (np.abs(b_lr).max() + 1e-12), abs(a_lm - b_lm) / (abs(b_lm) + 1e-12),
             np.abs(a_dp - b_dp).max() / (np.abs(b_dp).max() + 1e-12)]
        ok = max(e) < 2e-4 and np.all(a_dp[:, nout:] == 0)
        print(("ok   " if ok else "BAD  ") + "loss cfg %d: nout %d B %d  rows %.1e mean %.1e grad %.1e" % (seed0 + it, nout, B, *e), flush=True)
        bad += 0 if ok else 1
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = moves(n, s0) + slices(n, s0) + loss(n, s0)
    print("fuzz moves/loss: %d bad" % bad)
    sys.exit(1 if bad else 0)
