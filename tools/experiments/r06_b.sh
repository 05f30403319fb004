#!/bin/bash
# round 6, session B: sweep generator candidates without label noise (soft class overlap through coarse noise), 5 tasks each
set -u
mkdir -p gpurun_out
timeout 1500 python tools/experiments/r06_sweep_tune.py 5 0.25,1,8,5 0.25,1,8,7 0.25,1,8,10 0.25,1,8,14 0.25,1,8,20 0.25,0.98,8,5 1,1,8,10 > gpurun_out/r06_b_tune.txt 2> gpurun_out/r06_b_tune.err
tail -5 gpurun_out/r06_b_tune.err
cat gpurun_out/r06_b_tune.txt
