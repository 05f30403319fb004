#!/bin/bash
# round 6, second session, call 10: where the NaN of the closing run's sweep comes from
set -u
timeout 900 python -m pytest tests/test_gpu_sweep_paths.py -x -q -k three_kernel 2>&1 | grep -E "Error|assert|error|FAILED|passed|failed" | cut -c1-600 | head -20
echo "== main"; timeout 900 python tools/experiments/nan_diag.py 4 2>&1 | grep -v amdgpu.ids | tail -25 | cut -c1-220
echo "== bsf64"; CLHIP_LIB=clsurvey_amd/libclhip_bsf64.so timeout 900 python tools/experiments/nan_diag.py 4 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-220
