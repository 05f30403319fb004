#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "conv2d_general or alexnet or full_size or g15" 2>&1 | grep -E "passed|failed|Error|rel err|^FAILED|worst row" | head
echo "== alexnet step"; timeout 300 python tools/alexnet_step.py 128 10 2>&1 | tail -1
