#!/bin/bash
# round 6, session I: the free-running 10-task sweep (round 5's generator) + the teacher-forced comparison of the three kernel paths
set -u
mkdir -p gpurun_out
TUNE_FORCED=1 timeout 900 python tools/experiments/r06_sweep_tune.py 10 1,0.8,8,1,0,7 > gpurun_out/r06_i_tune.txt 2> gpurun_out/r06_i_tune.err
tail -15 gpurun_out/r06_i_tune.err
grep -v "omega max per" gpurun_out/r06_i_tune.txt
