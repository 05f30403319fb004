#!/bin/bash
for m in 2 0; do
  echo "== CLHIP_WGRAD_OVERLAP=$m"
  CLHIP_WGRAD_OVERLAP=$m timeout 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-configs --no-sweep 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
timeout 900 python -m pytest tests/test_gpu_framework.py tests/test_gpu_parity.py -x -q -k "g10 or g17 or g18 or determin or net or engine or full_size" 2>&1 | tail -3
