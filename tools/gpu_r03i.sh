#!/bin/bash
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_wino.py -q -p no:cacheprovider -x -k weight_gradient 2>&1 | tail -12
timeout 300 python tools/wino_bench.py 10 2>&1 | sed 's/fwd direct.*bwd_weight/bwd_weight/' | tail -10
for m in hat; do timeout 120 python tools/method_steps.py $m 224 6 2>&1 | tail -1; done
timeout 200 python -m pytest tests/test_gpu_wide.py tests/test_gpu_parity.py -q -p no:cacheprovider -x -k "hat or HAT" 2>&1 | tail -3
