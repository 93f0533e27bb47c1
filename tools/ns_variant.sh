#!/bin/bash
# Rebuild net_stream.hip with extra -D flags, run a command, restore the default build (GPU box).
flags="$1"; shift
touch linna_amd/csrc/net_stream.hip
LINNA_HIPCC_EXTRA="$flags" python linna_amd/_build.py > /dev/null 2>&1 || { echo build failed; exit 1; }
echo "== $flags"; "$@" 2>&1 | grep -v amdgpu
touch linna_amd/csrc/net_stream.hip; python linna_amd/_build.py > /dev/null 2>&1
