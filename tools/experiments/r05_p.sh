#!/bin/bash
set -u
mkdir -p gpurun_out/r05p; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_wino.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "weight_gradient or golden_g1 or full_size_vs_oracle or deterministic" 2>&1 | tail -3
timeout 200 python tools/conv_bench.py small 200 20 2>&1 | grep -v amdgpu.ids | grep "bwd_weight\|ALL" | cut -c1-30,100-140
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --no-sweep 2>/dev/null | cut -c1-220
