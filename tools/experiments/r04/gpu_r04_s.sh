#!/bin/bash
# round 4, call S: wino_wgrad_kernel (64 x 64 tiles: the wider nets' weight gradients) with 16-byte staging units (VEC, default) vs CLHIP_WG_VEC=0
set -u
mkdir -p gpurun_out/r04s; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_wino.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider \
    -k "weight_gradient or (full_size_vs_oracle and (base or wide_VGG9-8))" 2>&1 | tail -3
for v in vec scalar; do
  if [ $v = vec ]; then unset CLHIP_WG_VEC; else export CLHIP_WG_VEC=0; fi
  for w in wide base; do
    echo "== conv_bench $w $v"; timeout 200 python tools/conv_bench.py $w 200 20 2>&1 | tail -22 > gpurun_out/r04s/conv_${w}_$v.txt; grep "bwd_weight\|ALL" gpurun_out/r04s/conv_${w}_$v.txt | cut -c1-42,100-125
  done
done
