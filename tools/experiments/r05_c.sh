#!/bin/bash
# Round 5, GPU session C: bf16-split conv with weights staged through LDS — parity, per-layer timing (SEP on / off), step time.
set -u
mkdir -p gpurun_out/r05c; export TMPDIR=/tmp
O=gpurun_out/r05c
timeout 900 python -m pytest tests/test_gpu_bs.py -m gpu -x -q -p no:cacheprovider -s > $O/test_bs.txt 2>&1; echo "test_bs rc $?"; grep "error / sum" $O/test_bs.txt; tail -5 $O/test_bs.txt
for sep in 1 0; do
  CLHIP_BS_SEP=$sep timeout 300 python tools/bs_bench.py > $O/bs_bench_sep$sep.txt 2>&1; tail -16 $O/bs_bench_sep$sep.txt
done
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-configs --no-sweep"
timeout 300 $B > $O/bench_bs.txt 2> $O/bench_bs.err; tail -1 $O/bench_bs.txt | cut -c1-300
CLHIP_BS_SEP=0 timeout 300 $B > $O/bench_bs_sep0.txt 2> $O/bench_bs_sep0.err; tail -1 $O/bench_bs_sep0.txt | cut -c1-300
