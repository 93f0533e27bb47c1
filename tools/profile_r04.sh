#!/bin/bash
# GPU box: the round-4 profile set.  Summaries land in gpurun_out/prof_r04/ (copy what is to be judged into profiles/).
#  1. bench.py un-profiled (the bench line), kernel trace of the headline, FETCH_SIZE / WRITE_SIZE passes (tools/profile_bench.sh)
#  2. ONE workload per rocprofv3 run: ChtoModelv2(33,33) serving, ChtoModelv2(40,1000) dense serving, the training step at (26,457),
#     the one-call slice sampler at 4096 and 128 walkers, HMC transitions (MLP, ChtoModelv2)
#  3. matrix-pipe counters (separate --pmc passes): the headline kernel, ChtoModelv2(33,33) serving, the training step
set -e
root=$(pwd)
out=$root/gpurun_out/prof_r04
mkdir -p $out
tools/profile_bench.sh r04 > $out/profile_bench.log 2>&1 || { tail -20 $out/profile_bench.log; exit 1; }
echo "bench done"; tail -c 300 $out/bench.json; echo
tools/profile_cmd.sh r04_chto_v2 python tools/serve_probe.py ChtoModelv2 33 33 0 4096 2000 > $out/chto_v2.log 2>&1; tail -4 $out/chto_v2.log
tools/profile_cmd.sh r04_dense_1000 python tools/serve_probe.py ChtoModelv2 40 1000 1 4096 1000 > $out/dense_1000.log 2>&1; tail -4 $out/dense_1000.log
tools/profile_cmd.sh r04_training_26_457 python tools/train_probe.py 26 457 500 > $out/training.log 2>&1; tail -9 $out/training.log
SLICE_ONLY_FAST=1 tools/profile_cmd.sh r04_slice_4096 python tools/slice_probe.py 4096 > $out/slice_4096.log 2>&1; tail -9 $out/slice_4096.log
SLICE_ONLY_FAST=1 tools/profile_cmd.sh r04_slice_128 python tools/slice_probe.py 128 > $out/slice_128.log 2>&1; tail -9 $out/slice_128.log
tools/profile_cmd.sh r04_hmc_mlp python tools/hmc_probe.py MLP > $out/hmc_mlp.log 2>&1; tail -9 $out/hmc_mlp.log
tools/profile_cmd.sh r04_hmc_chto_v2 python tools/hmc_probe.py ChtoModelv2 > $out/hmc_v2.log 2>&1; tail -9 $out/hmc_v2.log
export TMPDIR=/tmp
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32"
cd /tmp
rocprofv3 --pmc $PMC --output-format csv -d $out/pmc_mfma_head -- python $root/tools/serve_probe.py MLP 33 33 0 4096 300 > $out/pmc_mfma_head.log 2>&1 || { tail -5 $out/pmc_mfma_head.log; exit 1; }
rocprofv3 --pmc $PMC --output-format csv -d $out/pmc_mfma_v2 -- python $root/tools/serve_probe.py ChtoModelv2 33 33 0 4096 300 > $out/pmc_mfma_v2.log 2>&1 || { tail -5 $out/pmc_mfma_v2.log; exit 1; }
rocprofv3 --pmc $PMC --output-format csv -d $out/pmc_mfma_train -- python $root/tools/train_probe.py 26 457 100 > $out/pmc_mfma_train.log 2>&1 || { tail -5 $out/pmc_mfma_train.log; exit 1; }
rocprofv3 --pmc $PMC --output-format csv -d $out/pmc_mfma_hmc_v2 -- python $root/tools/hmc_probe.py ChtoModelv2 > $out/pmc_mfma_hmc_v2.log 2>&1 || { tail -5 $out/pmc_mfma_hmc_v2.log; exit 1; }
cd $root
for t in head v2 train hmc_v2; do
  find $out/pmc_mfma_$t -name "*counter_collection.csv" | head -1 | xargs -I{} python tools/pmc_mfma.py {} $out/r04_${t}_pmc_mfma.json
  rm -rf $out/pmc_mfma_$t
done
