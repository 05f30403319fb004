#!/bin/bash
# round 4: ONE FULL TASK of the sweep on the host cores (5 gloo ranks x 16 threads) next to the GPU (tools/cpu_full_task.py)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import os; a=sorted(os.sched_getaffinity(0)); print('allowed cpus', len(a), a[:4], a[-4:])"
SECONDS=0
timeout 1700 python tools/cpu_full_task.py --ranks 5 --threads 16 --out gpurun_out/r04_cpu_full_task.json 2> gpurun_out/r04_cpu_full_task.err | cut -c1-3000
echo "cpu_full_task: $SECONDS s"; tail -25 gpurun_out/r04_cpu_full_task.err | cut -c1-300
