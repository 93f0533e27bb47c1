"""Timeline of the training step's launches from a rocprofv3 kernel trace (the chip-wide picture of VERDICT r5 item 2):
   cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -- python <repo>/tools/train_probe.py 26 457 200
   python tools/train_timeline.py <dir>
Per step (one merged launch `net_stream_kernel<6, 0, true, 3, 4>` each): when every launch of the step started and ended relative
to the merged launch's start, in microseconds -- medians over the steps of the trace's second half."""
import csv, glob, os, sys
import numpy as np
d = sys.argv[1]
f = [p for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)][0]
rows = list(csv.DictReader(open(f)))
ev = []
for r in rows:
    name = r["Kernel_Name"]
    short = ("chain" if "true, 3, 4>" in name else "dW" if "gemm_group_update" in name else "gate" if "pub_gate" in name else None)
    if short:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short))
ev.sort()
chains = [e for e in ev if e[2] == "chain"]
steps = []
for i in range(len(chains) // 2, len(chains) - 1):
    t0, t1 = chains[i][0], chains[i + 1][0]
    steps.append([(s - t0, e - t0, k) for s, e, k in ev if t0 <= s < t1])
n = max(len(s) for s in steps)
steps = [s for s in steps if len(s) == n]
print("%d steps of %d launches each; step period %.1f us (median)" % (len(steps), n, np.median(np.diff([c[0] for c in chains[len(chains) // 2:]])) / 1e3))
for j in range(n):
    kind = steps[0][j][2]
    st = np.median([s[j][0] for s in steps]) / 1e3
    en = np.median([s[j][1] for s in steps]) / 1e3
    print("  %-6s start %7.1f  end %7.1f  (%.1f us)" % (kind, st, en, en - st))
