#!/bin/bash
# Round 5, GPU session J: first layer on the bf16-split scheme (bs_c3_pool_kernel), three blocks per CU for the bs conv kernels.
set -u
mkdir -p gpurun_out/r05j; export TMPDIR=/tmp
O=gpurun_out/r05j
timeout 900 python -m pytest tests/test_gpu_bs.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "not engine_parity_with" > $O/test_sub.txt 2>&1; echo "tests rc $?"; tail -6 $O/test_sub.txt
timeout 200 python tools/conv_bench.py small 200 20 2>&1 | grep -v amdgpu.ids | tee $O/conv_layers_small.txt
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-configs --no-sweep"
timeout 300 $B > $O/bench.txt 2> $O/bench.err; tail -1 $O/bench.txt | cut -c1-300
CLHIP_BS=0 timeout 300 $B > $O/bench_bs0.txt 2> $O/bench_bs0.err; tail -1 $O/bench_bs0.txt | cut -c1-300
