#!/bin/bash
# Rebuild stream_mlp.hip with extra -D flags, run a command, restore the default build (GPU box).
# usage: tools/stream_variant.sh "-DSM_R=8" python tools/fused_bench.py
flags="$1"; shift
touch linna_amd/csrc/stream_mlp.hip
LINNA_HIPCC_EXTRA="$flags" python linna_amd/_build.py > /dev/null 2>&1 || { echo build failed; exit 1; }
echo "== $flags"; "$@" 2>&1 | grep -v amdgpu
touch linna_amd/csrc/stream_mlp.hip; python linna_amd/_build.py > /dev/null 2>&1
