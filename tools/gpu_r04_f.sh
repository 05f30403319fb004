#!/bin/bash
# round 4, call F: the hat224 fixture test (verbose), then ONE FULL TASK of the sweep on the host cores next to the GPU
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_wide.py -m gpu -q -p no:cacheprovider --tb=short -k "hat_step_wide" 2>&1 | grep -v "^$" | tail -15 | cut -c1-300
SECONDS=0
timeout 1500 python tools/cpu_full_task.py --ranks 5 --threads 16 --out gpurun_out/r04_cpu_full_task.json 2> gpurun_out/r04_cpu_full_task.err | cut -c1-2500
echo "cpu_full_task: $SECONDS s"; tail -3 gpurun_out/r04_cpu_full_task.err | cut -c1-300
