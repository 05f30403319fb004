#!/bin/bash
# round 6, session L: space-to-depth first layer of AlexNet (csrc/s2dconv.hip): parity, the AlexNet step with / without it, kernel split
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_s2d.py -m gpu -q -x -p no:cacheprovider --durations=5 2>&1 | tail -25 | tee gpurun_out/r06_l_tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_framework.py -m gpu -q -x -p no:cacheprovider -k "alexnet or conv2d or gem" 2>&1 | tail -8 | tee -a gpurun_out/r06_l_tests.log
echo "== alexnet step, s2d"; timeout 300 python tools/alexnet_step.py 128 10 2>&1 | tail -1 | tee gpurun_out/r06_l_step.log
echo "== alexnet step, CLHIP_S2D=0"; CLHIP_S2D=0 timeout 300 python tools/alexnet_step.py 128 10 2>&1 | tail -1 | tee -a gpurun_out/r06_l_step.log
bash tools/gpu_alex.sh r06_l_alex 2>&1 | tail -22 | tee gpurun_out/r06_l_alex.log
