#!/bin/bash
# GPU box: re-take what a late change touched: the bench line, and per argument one workload's kernel trace + matrix-pipe
# counters.  usage: tools/r04_refresh.sh [hmc_v2] [chto_v2]
set -e
root=$(pwd); out=$root/gpurun_out/prof_r04; mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
export TMPDIR=/tmp
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32"
for w in "$@"; do
  case $w in
    hmc_v2) tag=r04_hmc_chto_v2; cmd="tools/hmc_probe.py ChtoModelv2"; pm=hmc_v2;;
    chto_v2) tag=r04_chto_v2; cmd="tools/serve_probe.py ChtoModelv2 33 33 0 4096 2000"; pm=v2;;
    dense_1000) tag=r04_dense_1000; cmd="tools/serve_probe.py ChtoModelv2 40 1000 1 4096 1000"; pm=dense;;
    *) echo "unknown workload $w"; exit 1;;
  esac
  tools/profile_cmd.sh $tag python $cmd > $out/$w.log 2>&1; tail -3 $out/$w.log
  cd /tmp
  rocprofv3 --pmc $PMC --output-format csv -d $out/pmc_mfma_$pm -- python $root/${cmd%% *} ${cmd#* } > $out/pmc_mfma_$pm.log 2>&1 || { tail -5 $out/pmc_mfma_$pm.log; exit 1; }
  cd $root
  find $out/pmc_mfma_$pm -name "*counter_collection.csv" | head -1 | xargs -I{} python tools/pmc_mfma.py {} $out/r04_${pm}_pmc_mfma.json
  rm -rf $out/pmc_mfma_$pm
done
