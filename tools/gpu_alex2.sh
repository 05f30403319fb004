#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "alexnet or conv3x3 or conv2d or odd" 2>&1 | tail -3
bash tools/gpu_alex.sh ${1:-alex1}
