#!/bin/bash
# round 6, third session, call 2: weight operands of bs_conv_kernel two taps ahead (-DBS_PF=2, libclhip_pf2.so) against one tap ahead
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
CLHIP_LIB=$PWD/clsurvey_amd/libclhip_pf2.so timeout 1200 python -m pytest tests/test_gpu_bs.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/r06c2_tests.log
for rep in 1 2; do
for lib in libclhip.so libclhip_pf2.so; do
  for shape in "200 64 64 32" "192 64 64 32" "200 64 128 32" "200 64 64 16" "200 128 128 16"; do
    CLHIP_LIB=$PWD/clsurvey_amd/$lib timeout 120 python tools/bs_layer.py $shape 2>&1 | tail -1
  done
done
done | tee gpurun_out/r06c2_pf.txt
