#!/bin/bash
# round 4, call T (the last 0.8 GPU-minutes): wino_conv16_kernel with its A operands straight from L2 (CLHIP_W16_ADIRECT=1, two blocks per CU by
# registers) — parity on the 8 x 8 shapes and the per-layer times of small_VGG9 (shipped build in the same table: profiles/r04_last_bench_step.json)
set -u
mkdir -p gpurun_out/r04t; export TMPDIR=/tmp
export CLHIP_LIB=$PWD/clsurvey_amd/libclhip_w16ad.so
timeout 40 python -m pytest tests/test_gpu_wino.py -m gpu -q -x -p no:cacheprovider -k "data and (shape2] or shape3] or shape8] or shape12] or shape13] or shape14])" 2>&1 | tail -2
timeout 30 python tools/conv_bench.py small 200 20 2>&1 | tail -22 > gpurun_out/r04t/conv_small_w16ad.txt; grep "@8 \|ALL" gpurun_out/r04t/conv_small_w16ad.txt | cut -c1-42,100-125
