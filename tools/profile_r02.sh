#!/bin/bash
# GPU box: the round-2 profile set.  Summaries land in gpurun_out/prof_r02/ (copy what is to be judged into profiles/).
#  1. bench.py un-profiled (the bench line), kernel trace of bench.py, FETCH_SIZE / WRITE_SIZE passes (tools/profile_bench.sh)
#  2. ONE workload per rocprofv3 run: ChtoModelv2(33,33) serving, ChtoModelv2(40,1000) dense serving, the training step at (26,457)
#  3. matrix-pipe counters of the training step (separate --pmc pass)
set -e
root=$(pwd)
out=$root/gpurun_out/prof_r02
mkdir -p $out
tools/profile_bench.sh r02 > $out/profile_bench.log 2>&1 || { tail -20 $out/profile_bench.log; exit 1; }
cp gpurun_out/prof_r02/bench.json $out/bench.json 2>/dev/null || true
echo "bench done"; tail -c 400 $out/bench.json; echo
tools/profile_cmd.sh r02_chto_v2 python tools/serve_probe.py ChtoModelv2 33 33 0 4096 2000 > $out/chto_v2.log 2>&1; tail -4 $out/chto_v2.log
tools/profile_cmd.sh r02_dense_1000 python tools/serve_probe.py ChtoModelv2 40 1000 1 4096 1000 > $out/dense_1000.log 2>&1; tail -4 $out/dense_1000.log
tools/profile_cmd.sh r02_training_26_457 python tools/train_probe.py 26 457 500 > $out/training.log 2>&1; tail -9 $out/training.log
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $out/pmc_mfma -- python $root/tools/train_probe.py 26 457 100 > $out/pmc_mfma.log 2>&1 || { tail -5 $out/pmc_mfma.log; exit 1; }
cd $root
find $out/pmc_mfma -name "*counter_collection.csv" | head -1 | xargs -I{} python tools/pmc_mfma.py {} $out/r02_training_pmc_mfma.json
rm -rf $out/pmc_mfma
