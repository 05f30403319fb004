#!/bin/bash
# round 6, session A: candidates for the sweep's task generator (tools/experiments/r06_sweep_tune.py), 4 tasks each
set -u
mkdir -p gpurun_out
timeout 1500 python tools/experiments/r06_sweep_tune.py 4 1,0.8,8,1 0.5,0.8,8,1 0.25,0.8,8,1 0.125,0.8,8,1 0.25,0.9,8,1 0.25,1.0,8,3 0.25,0.8,4,1 0.0625,0.8,8,1 > gpurun_out/r06_a_tune.txt 2> gpurun_out/r06_a_tune.err
tail -5 gpurun_out/r06_a_tune.err
cat gpurun_out/r06_a_tune.txt
