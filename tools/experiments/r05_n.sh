#!/bin/bash
set -u
mkdir -p gpurun_out/r05n; export TMPDIR=/tmp
P=$PWD
for rep in 1 2; do
for shape in "200 64 64 32" "50 256 512 28"; do
  python tools/bs_layer.py $shape
  for v in 1 2 3; do CLHIP_LIB=$P/clsurvey_amd/libclhip_bsvar$v.so python tools/bs_layer.py $shape; done
done
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05n/variants.txt
