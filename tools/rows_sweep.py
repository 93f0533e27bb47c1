#!/usr/bin/env python
"""Whole-network kernel: time per launch against batch size for its three engines (16 rows per workgroup on
v_mfma_f32_16x16x4_f32; 8 and 4 rows on v_mfma_f32_4x4x1_16b_f32), forced through LINNA_NS_ROWS, and the
engine the batch size selects.  Evaluation (MLP 4x512 and ChtoModelv2 (33,33)), gradient (MLP) and the
training forward (ChtoModelv2, batch 500)."""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_paths import problem, timeit
from linna_amd import _lib

out = {}
for kind, kw in (("MLP", dict(width=512, depth=4)), ("ChtoModelv2", {})):
    p = problem(kind, 33, 33, False, **kw)
    lp = p["lp"]
    for B in (16, 64, 128, 256, 512, 1024, 2048, 4096):
        z = torch.randn(B, 33, device="cuda"); o = torch.empty(B, device="cuda"); g = torch.empty(B, 33, device="cuda")
        row = {}
        for rows in ("16", "8", "4", ""):
            if rows:
                _lib.engine_rows(int(rows))
            else:
                _lib.engine_rows(0)
            row["eval_" + (rows or "auto")] = round(timeit(lambda: lp.evaluate(z, out=o), 300) * 1e6, 1)
            if kind == "MLP":
                row["grad_" + (rows or "auto")] = round(timeit(lambda: lp.evaluate_with_grad(z, out=o, grad=g), 200) * 1e6, 1)
        out["%s B=%d" % (kind, B)] = row
        print(kind, B, row, flush=True)
    if kind == "ChtoModelv2":
        m = p["model"]
        x = torch.randn(500, 33, device="cuda")
        row = {}
        for rows in ("16", "8", "4", ""):
            if rows:
                _lib.engine_rows(int(rows))
            else:
                _lib.engine_rows(0)
            row["fwd500_" + (rows or "auto")] = round(timeit(lambda: m.forward(x), 300) * 1e6, 1)
        out["ChtoModelv2 training forward B=500"] = row
        print("train fwd", row, flush=True)
print(json.dumps(out))
