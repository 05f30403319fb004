#!/bin/bash
# round 6, session H: the ceiling regime with the LR grid restricted to its small values (every attempt below the stability limit),
# on the three kernel paths
set -u
mkdir -p gpurun_out
: > gpurun_out/r06_h_tune.txt
for grid in 5e-4,1e-4 1e-3,5e-4,1e-4; do
for bs in 1 0 2; do
  TUNE_LR_GRID=$grid CLHIP_BS=$bs timeout 600 python tools/experiments/r06_sweep_tune.py 10 0.25,0.9,8,1,0,7 >> gpurun_out/r06_h_tune.txt 2> gpurun_out/r06_h_tune.err
done
done
tail -5 gpurun_out/r06_h_tune.err
grep -v "omega max per" gpurun_out/r06_h_tune.txt
