#!/bin/bash
# round 6, third session, call 1: small-map geometry of bs_conv_kernel (64-pixel blocks of 32 x 32 wave tiles, four+ blocks per CU,
# weight operands two taps ahead) against the 128-pixel blocks: parity (tests/test_gpu_bs.py), per-layer timing A/B
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_bs.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/r06c1_tests.log
echo "== CLHIP_BS_SMALL=1" | tee gpurun_out/r06c1_bs_bench.txt
CLHIP_BS_SMALL=1 timeout 600 python tools/bs_bench.py 200 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06c1_bs_bench.txt
echo "== CLHIP_BS_SMALL=0" | tee -a gpurun_out/r06c1_bs_bench.txt
CLHIP_BS_SMALL=0 timeout 600 python tools/bs_bench.py 200 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06c1_bs_bench.txt
