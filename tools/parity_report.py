#!/usr/bin/env python
"""Fold a LINNA_PARITY_REPORT file (tests/parity.py) into one table: per call site the tolerance asserted and the worst
error measured over every test / parameter that reached it.

    LINNA_PARITY_REPORT=gpurun_out/parity.jsonl python -m pytest tests -m gpu -q
    python tools/parity_report.py gpurun_out/parity.jsonl [out.json]
"""
import json
import sys


def main():
    rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
    by = {}
    for r in rows:
        k = (r["site"], r["kind"], r["rtol"])              # (atol often scales with the data: the worst fraction of the tolerance is what counts)
        b = by.setdefault(k, dict(site=r["site"], kind=r["kind"], rtol=r["rtol"], atol=r["atol"], calls=0, worst_frac_of_tol=0.0,
                                  max_rel_err=0.0, max_abs_err=0.0, tests=set()))
        b["calls"] += 1
        b["atol"] = max(b["atol"], r["atol"])
        for f in ("worst_frac_of_tol", "max_rel_err", "max_abs_err"):
            b[f] = max(b[f], r[f])
        b["tests"].add(r["test"].split("::")[-1].split("[")[0])
    out = []
    for k in sorted(by, key=lambda k: (k[0].split(":")[0], int(k[0].split(":")[1]))):
        b = by[k]
        b["tests"] = sorted(b["tests"])
        b["measured_in_tol_metric"] = b["worst_frac_of_tol"] * b["rtol"] if b["rtol"] > 0 else b["max_abs_err"]
        out.append(b)
        print("%-28s %-8s rtol %-8.1e atol %-8.1e worst/tol %-9.3g measured %-9.3g %s" % (
            b["site"], b["kind"], b["rtol"], b["atol"], b["worst_frac_of_tol"], b["measured_in_tol_metric"], ",".join(b["tests"])[:60]))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
