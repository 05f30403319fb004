#!/bin/bash
# round 6, session P: bswgrad build 3 (explicit prologue, four rows per trip with renamed buffers): parity + timing
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_bswgrad.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/r06_p_tests.log
timeout 600 python tools/bs_wgrad_bench.py 200 64 64 32 200 64 64 16 200 64 128 16 200 128 128 16 200 128 256 16 200 256 256 16 200 64 128 32 2>&1 | tail -10 | tee gpurun_out/r06_p_bench.txt
