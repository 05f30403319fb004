#!/bin/bash
# round 4, call O: wino_conv16g_kernel with its A operands loaded from L2 straight into registers (CLHIP_W16G_ADIRECT=1: lane-ordered
# second image of U, LDS holds the halo planes only) vs the shipped build; `adirect_pu` adds the fenced burst on the UNPOOL instances
set -u
mkdir -p gpurun_out/r04o; export TMPDIR=/tmp
CLHIP_LIB=$PWD/clsurvey_amd/libclhip_adirect.so timeout 600 python -m pytest tests/test_gpu_wino.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider \
    -k "wino or golden_g1 or (full_size_vs_oracle and (small or base or wide_VGG9-8))" 2>&1 | tail -3
for v in default adirect adirect_pu; do
  if [ $v = default ]; then unset CLHIP_LIB; else export CLHIP_LIB=$PWD/clsurvey_amd/libclhip_$v.so; fi
  for w in small base wide; do
    echo "== conv_bench $w $v"; timeout 300 python tools/conv_bench.py $w 200 20 2>&1 | tail -22 > gpurun_out/r04o/conv_${w}_$v.txt; grep "ALL" gpurun_out/r04o/conv_${w}_$v.txt
  done
  echo "== bench step $v"; timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --no-sweep 2>/dev/null | cut -c1-200
done
