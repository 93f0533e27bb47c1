#!/usr/bin/env python
"""Time linna_logprob_eval (whole-network kernel) for the bench problem; LINNA_DISABLE_FUSED=1
selects the layer-by-layer path."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from linna_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lp, model, consts = bench.build_problem(torch.device("cuda", 0))
z = torch.randn(B, 33, device="cuda"); out = torch.empty(B, device="cuda")
for _ in range(20): lp.evaluate(z, out=out)
torch.cuda.synchronize()
e0, e1 = C.c_void_p(), C.c_void_p()
_lib.call("linna_event_create", C.byref(e0)); _lib.call("linna_event_create", C.byref(e1))
st = _lib.stream(); n = 200
_lib.call("linna_event_record", e0, st)
for _ in range(n): lp.evaluate(z, out=out)
_lib.call("linna_event_record", e1, st)
ms = C.c_float(); _lib.call("linna_event_elapsed_ms", e0, e1, C.byref(ms))
print("B=%d fused_disabled=%s  %.1f us/step" % (B, os.environ.get("LINNA_DISABLE_FUSED", "0"), ms.value / n * 1e3))
