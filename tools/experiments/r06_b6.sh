#!/bin/bash
# round 6, second session, call 6: first-layer weight gradient on the LDS-staged build — bias sums on the matrix pipe (u3mfma) and the
# block count of the launch (CLHIP_WG_SMALLC_BLOCKS)
set -u
for v in main u3mfma main u3mfma; do
  L=clsurvey_amd/libclhip_$v.so; [ $v = main ] && L=clsurvey_amd/libclhip.so
  echo -n "$v: "; CLHIP_LIB=$L timeout 120 python tools/conv_bench.py small 200 20 2>&1 | grep -E "bwd_weight_unpool +3x64@64|^ALL" | tr '\n' ' '; echo
done
for t in 512 768 1024 1536 2048 3072; do
  echo -n "blocks $t: "; CLHIP_WG_SMALLC_BLOCKS=$t timeout 120 python tools/conv_bench.py small 200 20 2>&1 | grep -E "bwd_weight_unpool +3x64@64|^ALL" | tr '\n' ' '; echo
done
CLHIP_LIB=clsurvey_amd/libclhip_u3mfma.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "wgrad or engine_matches or full_size" 2>&1 | tail -1
