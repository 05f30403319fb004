#!/bin/bash
# round 4, call R: wino_wgrad_ps_kernel staging in 16-byte pieces (VEC, the build's default where W % tile width == 0) vs CLHIP_WGPS_VEC=0
set -u
mkdir -p gpurun_out/r04r; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_wino.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider \
    -k "weight_gradient or golden_g1 or (full_size_vs_oracle and (small or base))" 2>&1 | tail -3
for v in vec scalar; do
  if [ $v = vec ]; then unset CLHIP_WGPS_VEC; else export CLHIP_WGPS_VEC=0; fi
  for w in small base; do
    echo "== conv_bench $w $v"; timeout 300 python tools/conv_bench.py $w 200 20 2>&1 | tail -22 > gpurun_out/r04r/conv_${w}_$v.txt; grep "bwd_weight\|ALL" gpurun_out/r04r/conv_${w}_$v.txt | cut -c1-42,100-125
  done
  echo "== bench step $v"; timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --no-sweep 2>/dev/null | cut -c1-200
done
