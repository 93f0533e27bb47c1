#!/bin/bash
# GPU box, round 4: the register-direct parameter-gradient launch (gemm.hip dw_body) -- parity suites in the default mode
# and on 64 x 64 tiles, then the training step of the bench under the kernel trace for the three K loops
# (LINNA_DW_DIRECT = 0 LDS-DMA ring, 1 direct 64 x 64, 2 direct 64 x 32 half-batch).  A step that dies (not: fails) ends the script.
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
run() { "$@"; rc=$?; if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "step died with $rc: $*"; exit $rc; fi; return 0; }
run timeout -k 10 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_train100k.py tests/test_gpu_fuzz.py -x -q > $out/t_dw2.log 2>&1; tail -4 $out/t_dw2.log
LINNA_DW_DIRECT=1 run timeout -k 10 400 python -m pytest tests/test_gpu_training.py -x -q > $out/t_dw1.log 2>&1; tail -4 $out/t_dw1.log
run timeout -k 10 300 python -m pytest tests/test_gpu_cond.py -q -s > $out/t_cond.log 2>&1; grep "^cond\|passed\|failed" $out/t_cond.log
for m in 0 1 2; do
  export LINNA_DW_DIRECT=$m
  run timeout -k 10 300 tools/profile_cmd.sh dw${m}_26_457 python tools/train_probe.py 26 457 500 > $out/dw${m}_26_457.log 2>&1; grep "us per step" $out/prof_dw${m}_26_457/trace.log; grep -i "group\|net_stream" $out/dw${m}_26_457.log | head -4
  run timeout -k 10 300 tools/profile_cmd.sh dw${m}_33_33 python tools/train_probe.py 33 33 500 > $out/dw${m}_33_33.log 2>&1; grep "us per step" $out/prof_dw${m}_33_33/trace.log; grep -i "group\|net_stream" $out/dw${m}_33_33.log | head -4
done
