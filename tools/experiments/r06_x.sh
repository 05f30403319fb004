#!/bin/bash
# round 6, session X: 5x5 weight gradient on the bf16-split scheme (csrc/bswgrad5.hip): parity + timing against the gather-GEMM
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_bswgrad.py -m gpu -q -x -p no:cacheprovider -k "5x5" 2>&1 | tail -6 | tee gpurun_out/r06_x_tests.log
timeout 300 python tools/bs_wgrad5_bench.py 128 64 192 27 2>&1 | tail -2 | tee gpurun_out/r06_x_bench.txt
