#!/usr/bin/env python
"""Randomised shapes / layouts / epilogues through linna_gemm_f32 against numpy (float64).
usage: fuzz_gemm.py [n] [seed0]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from linna_amd import _lib


def run(n, seed0):
    bad = 0
    ctx, st = _lib.ctx(), _lib.stream()
    for it in range(n):
        rs = np.random.RandomState(seed0 + it)
        M = int(rs.choice([1, 2, 15, 16, 33, 64, 65, 100, 128, 300, 500, 513, 1000, 4096]))
        N = int(rs.choice([1, 3, 16, 31, 33, 64, 65, 125, 250, 457, 500, 1000]))
        npairs = int(rs.choice([1, 1, 1, 2]))
        lays = [(0, 0), (0, 1), (1, 1)][int(rs.randint(0, 3))]
        aligned = bool(rs.randint(0, 4))                       # 1 in 4: row strides that are not multiples of 4 (slow path)
        pad = (lambda v: (v + 3) & ~3) if aligned else (lambda v: v + int(rs.randint(0, 3)))
        keep, ref = [], None
        g = _lib.Gemm(); g.npairs, g.M, g.N = npairs, M, N
        alpha0 = float(rs.choice([1.0, 0.1, -0.5])) if npairs == 2 or rs.randint(0, 2) else 1.0
        g.alpha0 = alpha0
        acc = np.zeros((M, N))
        for pi in range(npairs):
            K = int(rs.choice([1, 3, 4, 16, 31, 32, 33, 100, 128, 500, 512, 1016]))
            a = rs.standard_normal((M, K)); b = rs.standard_normal((N, K))
            if lays[0] == 0:
                A = torch.zeros(M, pad(K), device="cuda"); A[:, :K] = torch.as_tensor(a, dtype=torch.float32)
            else:
                A = torch.zeros(K, pad(M), device="cuda"); A[:, :M] = torch.as_tensor(a.T, dtype=torch.float32)
            if lays[1] == 0:
                Bm = torch.zeros(N, pad(K), device="cuda"); Bm[:, :K] = torch.as_tensor(b, dtype=torch.float32)
            else:
                Bm = torch.zeros(K, pad(N), device="cuda"); Bm[:, :N] = torch.as_tensor(b.T, dtype=torch.float32)
            keep += [A, Bm]
            g.p[pi].A, g.p[pi].lda, g.p[pi].alay = A.data_ptr(), A.stride(0), lays[0]
            g.p[pi].B, g.p[pi].ldb, g.p[pi].blay, g.p[pi].K = Bm.data_ptr(), Bm.stride(0), lays[1], K
            prod = a.astype(np.float32).astype(np.float64) @ b.astype(np.float32).astype(np.float64).T
            bias = None
            if rs.randint(0, 2):
                bias = torch.as_tensor(rs.standard_normal(N), dtype=torch.float32, device="cuda"); keep.append(bias)
                if pi == 0: g.bias0 = bias.data_ptr()
                else: g.bias1 = bias.data_ptr()
                prod = prod + bias.cpu().numpy().astype(np.float64)[None, :]
            acc = alpha0 * prod if pi == 0 and npairs == 1 else (alpha0 * prod if pi == 0 else acc + prod)
        v = acc
        if rs.randint(0, 3) == 0:
            R = torch.as_tensor(rs.standard_normal((M, pad(N))), dtype=torch.float32, device="cuda"); keep.append(R)
            g.R, g.ldr = R.data_ptr(), R.stride(0); v = v + R.cpu().numpy()[:, :N]
        if rs.randint(0, 2):
            g.relu = 1; v = np.maximum(v, 0)
        if rs.randint(0, 3) == 0:
            mk = torch.as_tensor(rs.standard_normal((M, pad(N))), dtype=torch.float32, device="cuda"); keep.append(mk)
            g.mask, g.ldmask = mk.data_ptr(), mk.stride(0); v = np.where(mk.cpu().numpy()[:, :N] > 0, v, 0)
        if rs.randint(0, 3) == 0:
            cs = torch.as_tensor(rs.uniform(0.5, 2, N), dtype=torch.float32, device="cuda"); ct = torch.as_tensor(rs.standard_normal(N), dtype=torch.float32, device="cuda")
            keep += [cs, ct]; g.cscale, g.cshift = cs.data_ptr(), ct.data_ptr(); v = v * cs.cpu().numpy()[None, :] + ct.cpu().numpy()[None, :]
        out = torch.full((M, (N + 3) & ~3), 3.0, device="cuda"); g.C, g.ldc = out.data_ptr(), out.stride(0)
        dot = rs.randint(0, 4) == 0
        if dot:
            dw = torch.as_tensor(rs.standard_normal((M, pad(N))), dtype=torch.float32, device="cuda"); keep.append(dw)
            slots = _lib.load().linna_gemm_dot_slots(M, N)
            part = torch.zeros(M, slots, device="cuda"); keep.append(part)
            g.dotwith, g.lddot, g.dot_partial, g.dot_slots = dw.data_ptr(), dw.stride(0), part.data_ptr(), slots
        tag = "gemm cfg %d: M %d N %d pairs %d K %s lay %s aligned %d relu %d R %d mask %d col %d dot %d" % (
            seed0 + it, M, N, npairs, [g.p[i].K for i in range(npairs)], lays, aligned, g.relu, bool(g.R), bool(g.mask), bool(g.cscale), dot)
        try:
            _lib.call("linna_gemm_f32", ctx, C.byref(g), st)
            torch.cuda.synchronize()
            got = out.cpu().numpy()[:, :N]
            scale = np.abs(v).max() + 1e-6
            Ksum = sum(g.p[i].K for i in range(npairs))
            e = np.abs(got - v).max() / scale
            ok = e < 2e-6 * max(8, Ksum ** 0.5) + 1e-6
            if dot:
                dref = np.sum(v * dw.cpu().numpy()[:, :N], axis=1)
                ed = np.abs(part.sum(1).cpu().numpy() - dref).max() / (np.abs(dref).max() + 1e-6)
                ok = ok and ed < 1e-4
            print(("ok   " if ok else "BAD  ") + tag + "  err %.1e" % e, flush=True)
            bad += 0 if ok else 1
        except Exception as ex:
            print("EXC  " + tag + "  " + repr(ex)[:200], flush=True); bad += 1
    print("fuzz gemm: %d configurations, %d bad" % (n, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
