#!/bin/bash
# round 6, session E: separable tasks with a LITTLE label noise (accuracies pinned at the data's ceiling, Fisher ~ flip rate), 10 tasks
set -u
mkdir -p gpurun_out
timeout 1700 python tools/experiments/r06_sweep_tune.py 10 0.25,0.99,8,1 0.25,0.98,8,1 0.25,0.995,8,1 1,0.99,8,1 1,0.98,8,1 0.25,0.97,8,1 > gpurun_out/r06_e_tune.txt 2> gpurun_out/r06_e_tune.err
tail -5 gpurun_out/r06_e_tune.err
cat gpurun_out/r06_e_tune.txt
