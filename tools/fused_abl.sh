#!/bin/bash
# Rebuild the fused kernel with each timing-only ablation and time it (GPU box).
for abl in "$@"; do
  touch linna_amd/csrc/fused_mlp.hip
  LINNA_HIPCC_EXTRA="-DFUSED_ABL=$abl" python linna_amd/_build.py > /dev/null 2>&1 || exit 1
  echo -n "ABL=$abl: "; python tools/fused_bench.py 2>&1 | grep us/step
done
touch linna_amd/csrc/fused_mlp.hip; python linna_amd/_build.py > /dev/null 2>&1
