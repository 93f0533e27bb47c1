"""The relation one step of emcee's stretch move (RedBlueMove + StretchMove, emcee 3.0.2) leaves between two
consecutive ensembles, as a checker shared by the CPU test on the reference-held emcee chain and the GPU test on the
HIP kernels' proposals.

A step shuffles the walkers into two halves; the first half moves against the second half's positions BEFORE the
step, the second against the first half's positions AFTER it.  A moved walker lies on the line through its old position
and ONE walker c of the complementary half:  new = c + zz (old - c),  zz in [1/a, a]."""
import itertools

import numpy as np


def partners(old, new, cands, a=2.0, tol=1e-9):
    """[(index, zz)] of the candidates ``(index, position)`` that explain ``old -> new``."""
    out = []
    for j, cj in cands:
        v0, v1 = old - cj, new - cj
        n0 = float(np.dot(v0, v0))
        if n0 == 0.0:
            continue
        zz = float(np.dot(v1, v0)) / n0
        if np.abs(v1 - zz * v0).max() <= tol * (1.0 + np.abs(cj).max()) and 1.0 / a - 1e-9 <= zz <= a + 1e-9:
            out.append((j, zz))
    return out


def explain_step(before, after, a=2.0, tol=1e-9):
    """A red/blue explanation of ``before[nw, nd] -> after[nw, nd]``: (first half, {moved walker: (partner, zz)}), or
    None if no equal split explains every moved walker."""
    nw = len(before)
    moved = [k for k in range(nw) if np.any(after[k] != before[k])]
    for A in itertools.combinations(range(nw), nw // 2):
        B = [k for k in range(nw) if k not in A]
        found = {}
        for k in moved:
            cands = [(j, before[j]) for j in B] if k in A else [(j, after[j]) for j in A]
            p = partners(before[k], after[k], cands, a, tol)
            if not p:
                break
            found[k] = p[0]
        else:
            return A, found
    return None
