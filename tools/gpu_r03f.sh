#!/bin/bash
export TMPDIR=/tmp
timeout 200 python tools/wino_bench.py 10 2>&1 | tail -10
echo "== bench (wino on)"; timeout 300 python bench.py --no-cpu-baseline --no-configs --no-sweep --steps 50 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'][:60], d['roofline']['frac'])"
echo "== bench (CLHIP_WINO=0)"; CLHIP_WINO=0 timeout 300 python bench.py --no-cpu-baseline --no-configs --no-sweep --steps 50 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
for m in packnet hat; do timeout 120 python tools/method_steps.py $m 64 10 2>&1 | tail -1; done
timeout 120 python tools/method_steps.py mas 64 10 2>&1 | tail -1
echo "== tests"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15
