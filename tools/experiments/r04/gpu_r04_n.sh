#!/bin/bash
# round 4, call N: fenced MFMA burst in wino_wgrad_ps_kernel (CLHIP_WGPS_FENCE=1) vs the shipped build
set -u
mkdir -p gpurun_out/r04n; export TMPDIR=/tmp
for v in default wgfence default; do
  if [ $v = default ]; then unset CLHIP_LIB; else export CLHIP_LIB=$PWD/clsurvey_amd/libclhip_$v.so; fi
  for w in small base; do
    echo "== conv_bench $w $v"; timeout 300 python tools/conv_bench.py $w 200 20 2>&1 | tail -22 > gpurun_out/r04n/conv_${w}_$v.txt; grep "bwd_weight\|ALL" gpurun_out/r04n/conv_${w}_$v.txt | cut -c1-30,100-125
  done
done
CLHIP_LIB=$PWD/clsurvey_amd/libclhip_wgfence.so timeout 600 python -m pytest tests/test_gpu_wino.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
