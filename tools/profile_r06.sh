#!/bin/bash
# GPU box: the round-6 profile set.  Summaries land in gpurun_out/prof_r06/ (copy what is to be judged into profiles/).
#  1. bench.py un-profiled (the bench line), kernel trace of the headline, FETCH_SIZE / WRITE_SIZE passes (tools/profile_bench.sh)
#  2. ONE workload per rocprofv3 run: ChtoModelv2(33,33) serving, ChtoModelv2(40,1000) dense serving (balanced triangular factor),
#     the training step at (26,457), whole training epochs, the emcee driver at 128 walkers (block entry of the stretch move +
#     the incremental autocorrelation kernels), the one-call slice sampler at 128 walkers, HMC transitions (ChtoModelv2)
#  3. matrix-pipe counters (separate --pmc passes): the headline kernel, ChtoModelv2(33,33) serving, dense_1000, the training step
# usage: tools/profile_r06.sh [part ...]   parts: bench work pmc (default: all)
set -e
root=$(pwd)
out=$root/gpurun_out/prof_r06
mkdir -p $out
parts="${@:-bench work pmc}"
for part in $parts; do
case $part in
bench)
  tools/profile_bench.sh r06 > $out/profile_bench.log 2>&1 || { tail -20 $out/profile_bench.log; exit 1; }
  echo "bench done"; tail -c 300 $out/bench.json; echo
  ;;
work)
  tools/profile_cmd.sh r06_chto_v2 python tools/serve_probe.py ChtoModelv2 33 33 0 4096 2000 > $out/chto_v2.log 2>&1; tail -4 $out/chto_v2.log
  tools/profile_cmd.sh r06_dense_1000 python tools/serve_probe.py ChtoModelv2 40 1000 1 4096 1000 > $out/dense_1000.log 2>&1; tail -4 $out/dense_1000.log
  tools/profile_cmd.sh r06_training_26_457 python tools/train_probe.py 26 457 500 > $out/training.log 2>&1; tail -6 $out/training.log
  tools/profile_cmd.sh r06_training_epochs python tools/epoch_bench.py 60 > $out/epochs.log 2>&1; tail -12 $out/epochs.log
  LINNA_PROBE_NSAMP=6000 tools/profile_cmd.sh r06_driver_128 python tools/driver_probe.py 128 > $out/driver_128.log 2>&1; tail -12 $out/driver_128.log
  SLICE_ONLY_FAST=1 tools/profile_cmd.sh r06_slice_128 python tools/slice_probe.py 128 > $out/slice_128.log 2>&1; tail -9 $out/slice_128.log
  tools/profile_cmd.sh r06_hmc_chto_v2 python tools/hmc_probe.py ChtoModelv2 > $out/hmc_v2.log 2>&1; tail -9 $out/hmc_v2.log
  ;;
pmc)
  export TMPDIR=/tmp
  PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32"
  cd /tmp
  rocprofv3 --pmc $PMC --output-format csv -d $out/pmc_mfma_head -- python $root/tools/serve_probe.py MLP 33 33 0 4096 300 > $out/pmc_mfma_head.log 2>&1 || { tail -5 $out/pmc_mfma_head.log; exit 1; }
  rocprofv3 --pmc $PMC --output-format csv -d $out/pmc_mfma_v2 -- python $root/tools/serve_probe.py ChtoModelv2 33 33 0 4096 300 > $out/pmc_mfma_v2.log 2>&1 || { tail -5 $out/pmc_mfma_v2.log; exit 1; }
  rocprofv3 --pmc $PMC --output-format csv -d $out/pmc_mfma_dense -- python $root/tools/serve_probe.py ChtoModelv2 40 1000 1 4096 200 > $out/pmc_mfma_dense.log 2>&1 || { tail -5 $out/pmc_mfma_dense.log; exit 1; }
  rocprofv3 --pmc $PMC --output-format csv -d $out/pmc_mfma_train -- python $root/tools/train_probe.py 26 457 100 > $out/pmc_mfma_train.log 2>&1 || { tail -5 $out/pmc_mfma_train.log; exit 1; }
  cd $root
  for t in head v2 dense train; do
    find $out/pmc_mfma_$t -name "*counter_collection.csv" | head -1 | xargs -I{} python tools/pmc_mfma.py {} $out/r06_${t}_pmc_mfma.json
    rm -rf $out/pmc_mfma_$t
  done
  ;;
esac
done
