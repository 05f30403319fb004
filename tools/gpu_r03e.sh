#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_wino.py -q -p no:cacheprovider -x 2>&1 | tail -8
timeout 300 python tools/wino_bench.py 10 2>&1 | tail -12
