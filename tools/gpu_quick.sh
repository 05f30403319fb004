#!/bin/bash
# parity subset + bench line (no cpu baseline): the inner loop of a tuning session
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider ${1:+-k "$1"} 2>&1 | tail -4
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tee gpurun_out/quick.json | cut -c1-330
