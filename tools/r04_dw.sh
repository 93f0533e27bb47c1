#!/bin/bash
# GPU box, round 4: K loops / tile shapes of the grouped parameter-gradient launch (gemm.hip) AS OF COMMIT 594ad1e (the variants
# were removed afterwards: at HEAD only mode 0 exists and the switch is ignored), LINNA_DW_DIRECT =
# 0 LDS-DMA ring on 64 x 64 tiles, 1 operands straight to registers on 64 x 64, 2 the same on 64 x 32 half-batch items,
# 3 LDS-DMA ring on 64 x 32 two-wave tiles.  usage: tools/r04_dw.sh "<modes to test>" "<modes to time>"
# A step that dies (not: fails) ends the script.
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
run() { "$@"; rc=$?; if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "step died with $rc: $*"; exit $rc; fi; return 0; }
for m in $1; do
  export LINNA_DW_DIRECT=$m
  run timeout -k 10 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_train100k.py tests/test_gpu_fuzz.py -x -q > $out/t_dw$m.log 2>&1; echo "mode $m: $(tail -1 $out/t_dw$m.log)"
done
for m in $2; do
  export LINNA_DW_DIRECT=$m
  run timeout -k 10 300 tools/profile_cmd.sh dw${m}_26_457 python tools/train_probe.py 26 457 500 > $out/dw${m}_26_457.log 2>&1; grep "us per step" $out/prof_dw${m}_26_457/trace.log; grep -i "group\|net_stream" $out/dw${m}_26_457.log | head -4
  run timeout -k 10 300 tools/profile_cmd.sh dw${m}_33_33 python tools/train_probe.py 33 33 500 > $out/dw${m}_33_33.log 2>&1; grep "us per step" $out/prof_dw${m}_33_33/trace.log; grep -i "group\|net_stream" $out/dw${m}_33_33.log | head -4
done
