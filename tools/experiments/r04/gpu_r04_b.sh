#!/bin/bash
# round 4, call B: new / tightened parity tests, the trajectory-separation test, stall counters, two sweeps (blobs q = 0.7 / 0.9)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04b
mkdir -p $O
echo "== new / tightened parity tests"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide.py -m gpu -q -p no:cacheprovider -s \
  -k "base_vgg9_widths or g34 or gem_gram or gem_qp or autograd_bridge or dropout_masks or ebll_step_with_dropout or si_golden or full_size or hat_step_wide or gem_observe" 2>&1 | grep -v "Warning\|warnings.warn\|^$" | tail -40 | tee $O/tests.txt
echo "== trajectory separation"
timeout 1200 python -m pytest tests/test_gpu_trajectory.py -m gpu -q -x -s -p no:cacheprovider 2>&1 | tail -30 | cut -c1-400 | tee $O/trajectory.txt
echo "== stall counters (small_VGG9)"
bash tools/gpu_stalls.sh small_VGG9_cl_128_128 r04b_stalls_small 2>&1 | tail -30
echo "== sweep q=0.7 (default)"
timeout 900 python bench.py --sweep-only --no-cpu-baseline 2> $O/sweep_q07.err > $O/sweep_q07.json; tail -2 $O/sweep_q07.err; cut -c1-2500 $O/sweep_q07.json
echo "== sweep q=0.9 + pair"
timeout 1200 python bench.py --sweep-only --sweep-blobs 8,4,1.2,0.9 2> $O/sweep_q09.err > $O/sweep_q09.json; tail -2 $O/sweep_q09.err; cut -c1-4000 $O/sweep_q09.json
