#!/bin/bash
# session X: weight-gradient block count of the merged grids on 16x16 maps (pixel splits of the 4 (k, c) tiles: 256 ... 800 blocks
# beside the 400 backward-data blocks; default 512): bench step per build
set -u
P=$PWD
for r in 1 2; do
  for lib in libclhip.so libclhip_t256.so libclhip_t384.so libclhip_t400.so libclhip_t624.so libclhip_t800.so; do
    CLHIP_LIB=$P/clsurvey_amd/$lib timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-sweep --no-configs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'ms_per_step %.4f' % d['ms_per_step'])"
  done
done
