#!/bin/bash
export TMPDIR=/tmp
echo "== bench (wino on)"; timeout 300 python bench.py --no-cpu-baseline --no-configs --no-sweep --steps 50 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
for m in packnet hat mas si; do timeout 120 python tools/method_steps.py $m 64 10 2>&1 | tail -1; done
for m in packnet hat; do timeout 120 python tools/method_steps.py $m 224 6 2>&1 | tail -1; done
echo "== tests"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -6
