#!/usr/bin/env python
"""Post-process a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
SQ_INSTS_VALU_MFMA_MOPS_F32` pass of bench.py into matrix-pipe utilisation per kernel variant
(rocprofv3's MfmaUtil expression: sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) x SIMDs); GRBM_GUI_ACTIVE is
reported summed over the 8 XCDs).  usage: pmc_mfma.py <counter_collection.csv> <out.json>"""
import collections, csv, json, sys

rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "net_stream_kernel" in k:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["duration_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    m["launches_averaged"] = len(d["GRBM_GUI_ACTIVE"])
    m["MfmaUtil_percent"] = 100.0 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4)
    m["mfma_flop_per_launch"] = m["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512.0
    out[k] = m
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, m in out.items():
    print("%-70s MfmaUtil %.1f %%  MFMA flop/launch %.3e  (%d launches, %.1f us under the profiler)" % (
        k[:70], m["MfmaUtil_percent"], m["mfma_flop_per_launch"], m["launches_averaged"], m["duration_ns"] / 1e3))
