#!/bin/bash
# session S: (1) upper bound of what one grid for backward-data + weight gradient of the small layers could give (the same launches on
# separate streams without events, tools/experiments/coresident_pair.py); (2) GEM kernels after the Gram kernel's 16-byte flag became a
# run-time argument (16 instances instead of 32)
set -u
mkdir -p gpurun_out/r05s; export TMPDIR=/tmp
timeout 200 python tools/experiments/coresident_pair.py 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05s/coresident_pair.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider --tb=short -k "gem" 2>&1 | tail -4 | cut -c1-300
