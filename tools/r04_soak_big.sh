#!/bin/bash
# GPU box: the long randomised soak at the round's final build (other seeds than tools/r04_soak.sh).  usage: r04_soak_big.sh net|rest
out=gpurun_out; mkdir -p $out
run() { "$@"; rc=$?; if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "step died with $rc: $*"; exit $rc; fi; return 0; }
if [ "$1" = "net" ]; then
  run timeout -k 10 1000 python tools/fuzz_net_stream.py 700 20000 > $out/soakbig_net.log 2>&1; tail -2 $out/soakbig_net.log
else
  run timeout -k 10 500 python tools/fuzz_train.py 350 20000 > $out/soakbig_train.log 2>&1; tail -1 $out/soakbig_train.log
  run timeout -k 10 300 python tools/fuzz_moves_loss.py 200 20000 > $out/soakbig_moves.log 2>&1; tail -1 $out/soakbig_moves.log
  run timeout -k 10 300 python tools/fuzz_gemm.py 1200 20000 > $out/soakbig_gemm.log 2>&1; tail -1 $out/soakbig_gemm.log
fi
