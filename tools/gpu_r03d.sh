#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_wino.py -q -p no:cacheprovider -x 2>&1 | tail -15
timeout 300 python tools/wino_bench.py 10 2>&1 | tail -12
timeout 120 python tools/method_steps.py hat 224 6 2>&1 | tail -1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/r03d_hat224 -- python $OLDPWD/tools/method_steps.py hat 224 6 > /dev/null 2>&1 )
f=$(find gpurun_out/r03d_hat224 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r03d_hat224_kernel_stats.csv && python tools/prof_stats.py gpurun_out/r03d_hat224_kernel_stats.csv 22
rm -rf gpurun_out/r03d_hat224
