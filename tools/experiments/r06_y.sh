#!/bin/bash
# round 6, session Y: the 5x5 bf16-split weight gradient in the plan executor: AlexNet parity tests, step with / without, GEM observe
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_framework.py tests/test_gpu_bswgrad.py tests/test_gpu_s2d.py -m gpu -q -x -p no:cacheprovider -k "alexnet or conv2d or gem or 5x5 or s2d or hat_alex" 2>&1 | tail -5 | tee gpurun_out/r06_y_tests.log
echo "== alexnet step" | tee gpurun_out/r06_y_step.txt; timeout 300 python tools/alexnet_step.py 128 10 2>&1 | tail -1 | tee -a gpurun_out/r06_y_step.txt
echo "== alexnet step CLHIP_BS_WGRAD=0" | tee -a gpurun_out/r06_y_step.txt; CLHIP_BS_WGRAD=0 timeout 300 python tools/alexnet_step.py 128 10 2>&1 | tail -1 | tee -a gpurun_out/r06_y_step.txt
bash tools/gpu_alex.sh r06_y_alex 2>&1 | tail -20 | tee gpurun_out/r06_y_alex.log
