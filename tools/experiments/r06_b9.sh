#!/bin/bash
# round 6, second session, call 9: first-layer forward with the B operands as 12 ds_read2_b64 (main) against 22 two-dword reads (fnw)
set -u
for v in fnw main fnw main; do
  L=clsurvey_amd/libclhip_$v.so; [ $v = main ] && L=clsurvey_amd/libclhip.so
  echo -n "$v: "; CLHIP_LIB=$L timeout 120 python tools/conv_bench.py small 200 20 2>&1 | grep -E "relu_pool_fwd +3x64@64|^ALL" | cut -c1-12,95-125 | tr '\n' ' '; echo
done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "fused_conv_relu_pool or engine_matches or full_size or g1" 2>&1 | tail -2
