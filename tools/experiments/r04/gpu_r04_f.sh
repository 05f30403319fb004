#!/bin/bash
# round 4: ONE FULL TASK of a sweep (every decision live: 5-LR grid, 70-epoch cap, stability decay) on the host cores (5 gloo
# ranks x 16 threads) next to the GPU, at a quarter of the sweep's task size so that the CPU side finishes in minutes
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
CLHIP_CPUTASK_SIZES=2000,500,500 timeout 700 python tools/cpu_full_task.py --ranks 5 --threads 16 --cpu-timeout 520 --out gpurun_out/r04_cpu_full_task.json 2> gpurun_out/r04_cpu_full_task.err | cut -c1-3000
echo "cpu_full_task: $SECONDS s"; tail -5 gpurun_out/r04_cpu_full_task.err | cut -c1-300
