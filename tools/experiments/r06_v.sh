#!/bin/bash
# round 6, session V: two-geometry launch of bs_conv_kernel (whole rounds of 128-pixel blocks + the last images as 64-pixel tiles): parity, timing
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_bs.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/r06_v_tests.log
for n in 192 200; do timeout 120 python tools/bs_layer.py $n 64 64 32 2>&1 | tail -1; done | tee gpurun_out/r06_v_mixed.txt
echo "-- CLHIP_BS_MIXED=0" | tee -a gpurun_out/r06_v_mixed.txt
for n in 200; do CLHIP_BS_MIXED=0 timeout 120 python tools/bs_layer.py $n 64 64 32 2>&1 | tail -1; done | tee -a gpurun_out/r06_v_mixed.txt
timeout 120 python tools/bs_layer.py 200 64 128 32 2>&1 | tail -1 | tee -a gpurun_out/r06_v_mixed.txt
CLHIP_BS_MIXED=0 timeout 120 python tools/bs_layer.py 200 64 128 32 2>&1 | tail -1 | tee -a gpurun_out/r06_v_mixed.txt
