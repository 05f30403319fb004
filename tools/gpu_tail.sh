#!/bin/bash
# fused classifier tail: bitwise tests, the wide-width goldens, a short bench, then the chaotic end-to-end fixtures
OUT=gpurun_out/${1:-tail}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_fc_tail.py tests/test_gpu_wide.py -x -q -s 2>&1 | tail -25 > $OUT/tests.log
timeout 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-configs --no-sweep > $OUT/bench.json 2> $OUT/bench.err
CLHIP_FC_TAIL=0 timeout 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-configs --no-sweep > $OUT/bench_notail.json 2>> $OUT/bench.err
timeout 900 python -m pytest tests/test_gpu_framework.py -x -q -k "g10 or g17 or g18 or g12 or g11" 2>&1 | tail -8 > $OUT/e2e.log
cat $OUT/tests.log; python - <<PY
import json
for f in ("bench", "bench_notail"):
    try:
        d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e: print(f, "failed", e)
PY
cat $OUT/e2e.log
