#!/bin/bash
# round 6, session C: 10-task sweeps of the stable-regime candidates (a little label noise, or large inputs that force small learning rates)
set -u
mkdir -p gpurun_out
timeout 1700 python tools/experiments/r06_sweep_tune.py 10 0.25,0.98,8,5 0.25,0.99,8,5 0.25,0.95,8,5 4,0.9,8,1 4,0.95,8,1 2,0.95,8,2 > gpurun_out/r06_c_tune.txt 2> gpurun_out/r06_c_tune.err
tail -5 gpurun_out/r06_c_tune.err
cat gpurun_out/r06_c_tune.txt
