for st in 0 4 8 16; do
  touch linna_amd/csrc/fused_mlp.hip
  LINNA_HIPCC_EXTRA="-DFUSED_STAGGER=$st" python linna_amd/_build.py > /dev/null 2>&1
  echo "STAGGER=$st $(python tools/fused_bench.py 2>&1 | tail -1)"
done
