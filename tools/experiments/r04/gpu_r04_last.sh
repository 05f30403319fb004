#!/bin/bash
# last GPU call of round 4 (1.8 GPU-minutes left): the conv / engine part of the GPU suite, smoke and the bench step on the final build
set -u
mkdir -p gpurun_out/r04last; export TMPDIR=/tmp
timeout 70 python -m pytest tests/test_gpu_wino.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "wino or golden_g1 or golden_g2 or golden_g5 or full_size_vs_oracle" 2>&1 | tail -2
timeout 30 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 40 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --no-sweep 2>/dev/null > gpurun_out/r04last/bench_step.json; cut -c1-220 gpurun_out/r04last/bench_step.json
