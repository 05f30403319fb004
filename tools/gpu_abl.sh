#!/bin/bash
# ablation timings of conv3x3.hip (fwd / bwd_data): variants built with CLHIP_ABL_* defines
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in default nogload nolstore nosync noepi nostage noall; do
  if [ $v = default ]; then unset CLHIP_LIB; else export CLHIP_LIB=$PWD/clsurvey_amd/libclhip_$v.so; fi
  echo "== $v"; timeout 200 python tools/conv_bench.py small 200 20 2>&1 | grep -v amdgpu.ids | tail -9
done 2>&1 | tee gpurun_out/abl.log
