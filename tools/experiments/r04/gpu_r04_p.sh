#!/bin/bash
# round 4, call P: halo planes of TWO chunks per LDS buffer (CLHIP_W16G_XPAIR=1: one block barrier per two chunks) vs the shipped build
# (A operands straight from L2, one barrier per chunk)
set -u
mkdir -p gpurun_out/r04p; export TMPDIR=/tmp
CLHIP_LIB=$PWD/clsurvey_amd/libclhip_xpair.so timeout 600 python -m pytest tests/test_gpu_wino.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider \
    -k "wino or golden_g1 or (full_size_vs_oracle and (small or base or wide_VGG9-8))" 2>&1 | tail -3
for v in default xpair; do
  if [ $v = default ]; then unset CLHIP_LIB; else export CLHIP_LIB=$PWD/clsurvey_amd/libclhip_$v.so; fi
  for w in small base wide; do
    echo "== conv_bench $w $v"; timeout 300 python tools/conv_bench.py $w 200 20 2>&1 | tail -22 > gpurun_out/r04p/conv_${w}_$v.txt; grep "ALL" gpurun_out/r04p/conv_${w}_$v.txt
  done
  echo "== bench step $v"; timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --no-sweep 2>/dev/null | cut -c1-200
done
