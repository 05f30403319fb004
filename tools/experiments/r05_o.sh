#!/bin/bash
# Round 5, GPU session O: block count (= pixel splits x (k, c) tiles) of the pixel-split Winograd weight gradient on the small layers
set -u
mkdir -p gpurun_out/r05o; export TMPDIR=/tmp
P=$PWD
for v in default 256 384 768 1024 default; do
  if [ $v = default ]; then unset CLHIP_LIB; else export CLHIP_LIB=$P/clsurvey_amd/libclhip_wgps$v.so; fi
  echo "== blocks $v"; timeout 120 python tools/conv_bench.py small 200 20 2>&1 | grep "bwd_weight\|ALL"
done 2>&1 | grep -v amdgpu.ids | cut -c1-30,100-140 | tee gpurun_out/r05o/wgps_blocks.txt
