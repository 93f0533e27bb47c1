#!/bin/bash
# GPU box: rocprofv3 kernel-trace summary + the two PMC passes of bench.py; results under gpurun_out/prof_<tag>/.
# usage: tools/profile_bench.sh <tag>      (then copy the summaries into profiles/)
set -e
tag=${1:-rXX}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
python bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
tail -c 600 $out/bench.json; echo
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $root/bench.py --no-cpu-baseline --no-secondary --no-driver --steps 100 > $out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python $root/bench.py --no-cpu-baseline --no-secondary --no-driver --steps 20 > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python $root/bench.py --no-cpu-baseline --no-secondary --no-driver --steps 20 > $out/pmc_write.log 2>&1
cd $root
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out/pmc_fetch -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $out/pmc_fetch_size.csv
find $out/pmc_write -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $out/pmc_write_size.csv
head -4 $out/kernel_stats.csv
rm -rf $out/trace $out/pmc_fetch $out/pmc_write
