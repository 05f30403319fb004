#!/bin/bash
# round 4, call C: re-run of the revised tests, trajectory separation, sweep at q = 0.8, the default bench line
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04c
mkdir -p $O
echo "== revised parity tests"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide.py -m gpu -q -p no:cacheprovider -s --tb=short \
  -k "base_vgg9_widths or g34 or gem_gram or full_size or hat_step_wide" > $O/tests.txt 2>&1; grep -v "Warning\|warnings.warn\|^$" $O/tests.txt | tail -45 | cut -c1-330
echo "== trajectory separation"
timeout 1500 python -m pytest tests/test_gpu_trajectory.py -m gpu -q -x -s -p no:cacheprovider --tb=short > $O/trajectory.txt 2>&1; tail -32 $O/trajectory.txt | cut -c1-420
echo "== sweep q=0.8 (GPU only)"
timeout 900 python bench.py --sweep-only --no-cpu-baseline --sweep-blobs 8,4,1.2,0.8 2> $O/sweep_q08.err > $O/sweep_q08.json; tail -2 $O/sweep_q08.err; cut -c1-1500 $O/sweep_q08.json
echo "== default bench"
/usr/bin/time -v timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; grep "Elapsed" $O/bench.err; tail -3 $O/bench.err | cut -c1-300; cut -c1-1200 $O/bench.json
