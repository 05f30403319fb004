#!/bin/bash
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_wino.py -q -p no:cacheprovider -x 2>&1 | tail -3
timeout 200 python tools/wino_bench.py 10 2>&1 | tail -10
echo "== bench (wino on)"; timeout 300 python bench.py --no-cpu-baseline --no-configs --no-sweep --steps 50 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
for m in packnet hat mas; do timeout 120 python tools/method_steps.py $m 64 10 2>&1 | tail -1; done
