#!/bin/bash
# round 6, second session, call 8: conv3x3_c3w64_relu_pool_kernel with the halo staging moved between the MFMAs and the output stores
# (main) against the build before this session (u3old); ablations 1 (no stores) and 8 (no MFMAs) of the new build; parity
set -u
for v in u3old main fa1 fa8 main u3old; do
  L=clsurvey_amd/libclhip_$v.so; [ $v = main ] && L=clsurvey_amd/libclhip.so
  echo -n "$v: "; CLHIP_LIB=$L timeout 120 python tools/conv_bench.py small 200 20 2>&1 | grep -E "relu_pool_fwd +3x64@64|bwd_weight_unpool +3x64@64|^ALL" | cut -c1-12,95-125 | tr '\n' ' '; echo
done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "fused_conv_relu_pool or engine_matches or full_size or g1 or wgrad" 2>&1 | tail -2
