#!/bin/bash
# round 6, session J: (1) the new batch-200 parity legs against torch CPU (bs / wino / pair / engine base N=200 / sweep paths);
# (2) A/B of the conflict-free LDS pitches of wino_wgrad_ps_kernel (libclhip_wgpsold.so = round 5's pitches)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== conv_bench small new"; timeout 300 python tools/conv_bench.py small 200 20 2>&1 | tail -24 | tee gpurun_out/r06_j_small_new.log
echo "== conv_bench small old pitches"; CLHIP_LIB=$PWD/clsurvey_amd/libclhip_wgpsold.so timeout 300 python tools/conv_bench.py small 200 20 2>&1 | tail -24 | tee gpurun_out/r06_j_small_old.log
echo "== conv_bench base new"; timeout 300 python tools/conv_bench.py base 200 10 2>&1 | tail -24 | tee gpurun_out/r06_j_base_new.log
echo "== conv_bench base old"; CLHIP_LIB=$PWD/clsurvey_amd/libclhip_wgpsold.so timeout 300 python tools/conv_bench.py base 200 10 2>&1 | tail -24 | tee gpurun_out/r06_j_base_old.log
echo "== tests"
timeout 2400 python -m pytest tests/test_gpu_bs.py tests/test_gpu_wino.py tests/test_gpu_pair.py tests/test_gpu_sweep_paths.py -m gpu -q -x -p no:cacheprovider --durations=8 2>&1 | tail -25 | tee gpurun_out/r06_j_tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "full_size" --durations=5 2>&1 | tail -12 | tee gpurun_out/r06_j_tests2.log
