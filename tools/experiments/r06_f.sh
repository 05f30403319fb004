#!/bin/bash
# round 6, session F: un-normalised inputs (large amplitude) so that the large learning rates of the grid fail in phase 1
set -u
mkdir -p gpurun_out
timeout 1700 python tools/experiments/r06_sweep_tune.py 4 8,0.9,8,1 16,0.9,8,1 32,0.9,8,1 64,0.9,8,1 128,0.9,8,1 > gpurun_out/r06_f_tune.txt 2> gpurun_out/r06_f_tune.err
tail -5 gpurun_out/r06_f_tune.err
cat gpurun_out/r06_f_tune.txt
