#!/bin/bash
# GPU box: rocprofv3 kernel-trace summary of ONE workload:  tools/profile_cmd.sh <tag> python tools/xyz.py args...
# (the program itself follows the tag: no env / bash -c hop under the profiler); summary -> gpurun_out/prof_<tag>_kernel_stats.csv
set -e
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
prog=$1; shift
script=$root/$1; shift
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $prog $script "$@" > $out/trace.log 2>&1 || { tail -20 $out/trace.log; exit 1; }
cd $root
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $root/gpurun_out/prof_${tag}_kernel_stats.csv
tail -3 $out/trace.log
rm -rf $out/trace
python - <<PY
import csv
rows = list(csv.DictReader(open("$root/gpurun_out/prof_${tag}_kernel_stats.csv")))
for r in rows[:16]:
    print("%-70s calls %6s  avg %9.2f us  total %6.2f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
