#!/bin/bash
# round 6, session Z: the tap-split kernel with 3 x 3 taps on AlexNet's 13 x 13 layers: parity + timing against the Winograd weight gradient
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_bswgrad.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/r06_z_tests.log
timeout 300 python tools/bs_wgrad_bench.py 128 192 384 13 128 384 256 13 128 256 256 13 2>&1 | tail -3 | tee gpurun_out/r06_z_bench.txt
timeout 300 python tools/bs_wgrad5_bench.py 128 64 192 27 2>&1 | tail -1 | tee -a gpurun_out/r06_z_bench.txt
