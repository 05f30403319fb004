#!/bin/bash
# full 10-task EWC sweep at Tiny-ImageNet shapes through the driver (tools/sweep.py); usage: gpu_sweep.sh [timeout_s]
set -u
mkdir -p gpurun_out
timeout ${1:-420} python tools/sweep.py --cpu-train-rate 527 2> gpurun_out/sweep.err | tail -1 | tee gpurun_out/sweep.json
tail -3 gpurun_out/sweep.err
