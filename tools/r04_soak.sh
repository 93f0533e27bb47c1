#!/bin/bash
# GPU box, round 4: the randomised parity soak after the round's kernel changes (dense log-likelihood as |d L|^2, two
# accumulator chains in the grouped parameter-gradient launch, single slot computation in the update epilogue).
out=gpurun_out; mkdir -p $out
run() { "$@"; rc=$?; if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "step died with $rc: $*"; exit $rc; fi; return 0; }
run timeout -k 10 500 python tools/fuzz_net_stream.py 300 4000 > $out/soak_net.log 2>&1; tail -3 $out/soak_net.log
run timeout -k 10 500 python tools/fuzz_train.py 250 4000 > $out/soak_train.log 2>&1; tail -3 $out/soak_train.log
run timeout -k 10 300 python tools/fuzz_moves_loss.py 120 4000 > $out/soak_moves.log 2>&1; tail -3 $out/soak_moves.log
run timeout -k 10 300 python tools/fuzz_gemm.py 500 4000 > $out/soak_gemm.log 2>&1; tail -3 $out/soak_gemm.log
