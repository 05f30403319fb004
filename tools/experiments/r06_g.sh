#!/bin/bash
# round 6, session G: seeds of the label-noise-ceiling regime — which sequences keep every stability-decay decision away from the limit?
set -u
mkdir -p gpurun_out
specs=""
for seed in 1 2 3 4 5 6 7 8; do specs="$specs 0.25,0.9,8,1,0,$seed"; done
for seed in 1 2 3 4 5 6 8; do specs="$specs 1,0.9,8,1,0,$seed"; done
timeout 1700 python tools/experiments/r06_sweep_tune.py 10 $specs > gpurun_out/r06_g_tune.txt 2> gpurun_out/r06_g_tune.err
tail -5 gpurun_out/r06_g_tune.err
grep -v "omega max per\|phase-1 grid" gpurun_out/r06_g_tune.txt
