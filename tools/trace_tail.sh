#!/bin/bash
# GPU box: rocprofv3 kernel trace of ONE workload, the last N dispatches listed in order (name, grid, duration):
#   tools/trace_tail.sh <tag> <N> python tools/xyz.py args...   -> gpurun_out/trace_<tag>.txt
set -e
tag=$1; shift
n=$1; shift
root=$(pwd)
out=$root/gpurun_out/trace_$tag
mkdir -p $out
export TMPDIR=/tmp
prog=$1; shift
script=$root/$1; shift
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- $prog $script "$@" > $out/trace.log 2>&1 || { tail -20 $out/trace.log; exit 1; }
cd $root
f=$(find $out -name "*kernel_trace.csv" | head -1)
python - "$f" "$n" > $root/gpurun_out/trace_$tag.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows[-int(sys.argv[2]):]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None:
        t0 = s
    print("%10.1f us  +%8.1f us  grid %8s wg %4s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), r["Kernel_Name"][:80]))
PY
rm -rf $out
tail -3 $root/gpurun_out/trace_$tag.txt
