#!/bin/bash
# round 6, session D: the same 10-task sweep on the three fp32-grade kernel paths (CLHIP_BS=0 Winograd f32, 1 default, 2 bf16-split
# wherever it runs) for three candidate task generators — how far apart do the paths end, task by task?
set -u
mkdir -p gpurun_out
: > gpurun_out/r06_d_tune.txt
for bs in 0 1 2; do
  CLHIP_BS=$bs timeout 900 python tools/experiments/r06_sweep_tune.py 10 0.25,1,8,5 0.25,1,8,3 0.25,0.99,8,5 >> gpurun_out/r06_d_tune.txt 2> gpurun_out/r06_d_tune.err
done
tail -5 gpurun_out/r06_d_tune.err
cat gpurun_out/r06_d_tune.txt
