#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "engine or alexnet or gem or golden" 2>&1 | tail -5
echo "== overlap on";  timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tee gpurun_out/ov_on.json | cut -c1-400
echo "== overlap off"; CLHIP_WGRAD_OVERLAP=0 timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tee gpurun_out/ov_off.json | cut -c1-400
echo "== alexnet on";  timeout 300 python tools/alexnet_step.py 128 10 2>&1 | tail -1
echo "== alexnet off"; CLHIP_WGRAD_OVERLAP=0 timeout 300 python tools/alexnet_step.py 128 10 2>&1 | tail -1
