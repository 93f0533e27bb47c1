#!/bin/bash
# GPU box: re-take what a late kernel change touched (the one-launch gradient): its kernel trace, its matrix-pipe counters,
# and the bench line.
set -e
root=$(pwd); out=$root/gpurun_out/prof_r04; mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
tools/profile_cmd.sh r04_hmc_chto_v2 python tools/hmc_probe.py ChtoModelv2 > $out/hmc_v2.log 2>&1; tail -4 $out/hmc_v2.log
export TMPDIR=/tmp
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32"
cd /tmp
rocprofv3 --pmc $PMC --output-format csv -d $out/pmc_mfma_hmc_v2 -- python $root/tools/hmc_probe.py ChtoModelv2 > $out/pmc_mfma_hmc_v2.log 2>&1 || { tail -5 $out/pmc_mfma_hmc_v2.log; exit 1; }
cd $root
find $out/pmc_mfma_hmc_v2 -name "*counter_collection.csv" | head -1 | xargs -I{} python tools/pmc_mfma.py {} $out/r04_hmc_v2_pmc_mfma.json
rm -rf $out/pmc_mfma_hmc_v2
