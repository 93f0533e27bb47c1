#!/bin/bash
# GPU box: phase stamps of the grouped parameter-gradient + AdamW launch (LDS-DMA ring, LINNA_DW_DIRECT=0)
python linna_amd/_build.py --stamps --source=gemm.hip -DGEMM_STAMPS > gpurun_out/stamps_build.log 2>&1 || { tail -5 gpurun_out/stamps_build.log; exit 1; }
LINNA_DW_DIRECT=0 timeout -k 10 300 python tools/gemm_stamps.py 26 457 500 > gpurun_out/gemm_stamps_26_457.log 2>&1; cat gpurun_out/gemm_stamps_26_457.log | tail -14
