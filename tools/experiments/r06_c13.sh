#!/bin/bash
# round 6, third session, call 13: Winograd U images and bf16-split images of a pass in ONE launch: parity (engine, bs, wino, pair, modes), step, kernel list of a pass
set -u
mkdir -p gpurun_out/r06c13; export TMPDIR=/tmp
timeout 1700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bs.py tests/test_gpu_wino.py tests/test_gpu_pair.py tests/test_gpu_modes.py tests/test_gpu_wide.py tests/test_gpu_switches.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/r06c13_tests.log
B="bench.py --no-cpu-baseline --no-configs --no-sweep --steps 200 --warmup 20"
for rep in 1 2 3; do timeout 300 python $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step ms %.4f' % d['ms_per_step'])"; done | tee gpurun_out/r06c13_step.txt
P=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $P/gpurun_out/r06c13/prof -- python $P/tools/one_step.py 24 small_VGG9_cl_128_128 > $P/gpurun_out/r06c13/prof.log 2>&1 )
f=$(find gpurun_out/r06c13/prof -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py "$f" | tee gpurun_out/r06c13_gaps.txt
rm -rf gpurun_out/r06c13/prof
timeout 200 python tools/alexnet_step.py 128 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/r06c13_step.txt
