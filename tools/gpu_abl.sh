#!/bin/bash
# ablation timings: variants built with build_variant(); usage: bash tools/gpu_abl.sh v1 v2 ...
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in default "$@"; do
  if [ $v = default ]; then unset CLHIP_LIB; else export CLHIP_LIB=$PWD/clsurvey_amd/libclhip_$v.so; fi
  echo "== $v"; timeout 200 python tools/conv_bench.py small 200 20 2>&1 | grep -v amdgpu.ids | grep "64x64  @32\|64x64  @16" | head -2
done 2>&1 | tee gpurun_out/abl.log
