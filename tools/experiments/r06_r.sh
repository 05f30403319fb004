#!/bin/bash
# round 6, session R: bswgrad in the plan executor: engine / wide / modes parity, per-layer timings per width, step A/B against CLHIP_BS_WGRAD=0
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide.py tests/test_gpu_bs.py tests/test_gpu_bswgrad.py tests/test_gpu_modes.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/r06_r_tests.log
for w in small base wide; do echo "== conv_bench $w"; timeout 300 python tools/conv_bench.py $w 200 10 2>&1 | grep "bwd_weight\|TOTAL\|ALL"; done | tee gpurun_out/r06_r_conv.txt
for w in small base wide; do echo "== conv_bench $w CLHIP_BS_WGRAD=0"; CLHIP_BS_WGRAD=0 timeout 300 python tools/conv_bench.py $w 200 10 2>&1 | grep "bwd_weight\|TOTAL\|ALL"; done | tee -a gpurun_out/r06_r_conv.txt
echo "== bench step (bs wgrad)"; timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-sweep --kernel-iters 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('configs_ms_per_step'))" | tee gpurun_out/r06_r_step.txt
echo "== bench step CLHIP_BS_WGRAD=0"; CLHIP_BS_WGRAD=0 timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-sweep --kernel-iters 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('configs_ms_per_step'))" | tee -a gpurun_out/r06_r_step.txt
