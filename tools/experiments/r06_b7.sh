#!/bin/bash
# round 6, second session, call 7: timing-only ablations of conv3x3_c3w64_relu_pool_kernel (C3W64_ABL: 1 no stores, 2 no pooling
# arithmetic, 4 no LDS reads of B, 8 no MFMAs, 7 = 1 + 2 + 4) and the build adopted for the first-layer weight gradient
set -u
for v in main fa1 fa2 fa4 fa8 fa7 main; do
  L=clsurvey_amd/libclhip_$v.so; [ $v = main ] && L=clsurvey_amd/libclhip.so
  echo -n "$v: "; CLHIP_LIB=$L timeout 120 python tools/conv_bench.py small 200 20 2>&1 | grep -E "relu_pool_fwd +3x64@64|bwd_weight_unpool +3x64@64" | cut -c1-12,95-125 | tr '\n' ' '; echo
done
