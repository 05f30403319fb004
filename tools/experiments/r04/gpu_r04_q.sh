#!/bin/bash
# round 4, call Q: with the round's kernels, (1) weight gradients on the side stream again (CLHIP_WGRAD_OVERLAP=2: layers of 16 x 16 and
# smaller; =1: every conv layer), (2) 8 x 8 layers on wino_conv16g_kernel<4, 4, 4> instead of wino_conv16_kernel (CLHIP_WINO16_BELOW_UNITS=0)
set -u
mkdir -p gpurun_out/r04q; export TMPDIR=/tmp
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --no-sweep"
echo "== default";  timeout 300 $B 2>/dev/null | cut -c1-200
echo "== overlap 2"; CLHIP_WGRAD_OVERLAP=2 timeout 300 $B 2>/dev/null | cut -c1-200
echo "== overlap 1"; CLHIP_WGRAD_OVERLAP=1 timeout 300 $B 2>/dev/null | cut -c1-200
export CLHIP_LIB=$PWD/clsurvey_amd/libclhip_no16.so
echo "== no16"; timeout 300 $B 2>/dev/null | cut -c1-200
timeout 300 python tools/conv_bench.py small 200 20 2>&1 | tail -22 > gpurun_out/r04q/conv_small_no16.txt; grep "@8 \|ALL" gpurun_out/r04q/conv_small_no16.txt | cut -c1-125
