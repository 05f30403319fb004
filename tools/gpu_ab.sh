#!/bin/bash
# quick A/B: parity tests for conv + conv_bench with/without an env switch
set -u
mkdir -p gpurun_out
TAG=${1:-ab}
SW=${2:-CLHIP_WGRAD_V1}
export TMPDIR=/tmp
echo "== tests (conv + engine)"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "conv or engine or golden" 2>&1 | tail -8
echo "== conv_bench new"; timeout 300 python tools/conv_bench.py small 200 20 2>&1 | tail -10 | tee gpurun_out/${TAG}_new.log
echo "== conv_bench $SW=1"; env $SW=1 timeout 300 python tools/conv_bench.py small 200 20 2>&1 | tail -10 | tee gpurun_out/${TAG}_old.log
echo "== conv_bench base (N=200)"; timeout 300 python tools/conv_bench.py base 200 10 2>&1 | tail -10 | tee gpurun_out/${TAG}_base.log
