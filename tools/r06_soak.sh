#!/bin/bash
# GPU box, round 6: the randomised parity soaks at the round's final build (slice kernels with zeus' budget / acceptance rule,
# ABI 11 descriptors, everything else unchanged), plus the slice fuzz (every fusion mask = mask 0 = the round loop).
out=gpurun_out; mkdir -p $out
run() { "$@"; rc=$?; if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "step died with $rc: $*"; exit $rc; fi; return 0; }
run timeout -k 10 300 python tools/fuzz_slice.py 60 9000 > $out/soak_slice.log 2>&1; tail -2 $out/soak_slice.log
run timeout -k 10 400 python tools/fuzz_net_stream.py 200 6000 > $out/soak_net.log 2>&1; tail -2 $out/soak_net.log
run timeout -k 10 400 python tools/fuzz_train.py 150 6000 > $out/soak_train.log 2>&1; tail -2 $out/soak_train.log
run timeout -k 10 300 python tools/fuzz_moves_loss.py 100 6000 > $out/soak_moves.log 2>&1; tail -2 $out/soak_moves.log
run timeout -k 10 200 python tools/fuzz_gemm.py 400 6000 > $out/soak_gemm.log 2>&1; tail -2 $out/soak_gemm.log
