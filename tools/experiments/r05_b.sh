#!/bin/bash
# Round 5, GPU session B: first run of the bf16-split conv kernels — parity tests, per-layer timing vs Winograd, step time.
set -u
mkdir -p gpurun_out/r05b; export TMPDIR=/tmp
O=gpurun_out/r05b
timeout 900 python -m pytest tests/test_gpu_bs.py -m gpu -x -q -p no:cacheprovider -s > $O/test_bs.txt 2>&1; echo "test_bs rc $?"; tail -15 $O/test_bs.txt
for bm in auto 128 256; do
  if [ $bm = auto ]; then unset CLHIP_BS_BM; else export CLHIP_BS_BM=$bm; fi
  timeout 300 python tools/bs_bench.py > $O/bs_bench_$bm.txt 2>&1; tail -17 $O/bs_bench_$bm.txt
done
unset CLHIP_BS_BM
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-configs --no-sweep"
timeout 300 $B > $O/bench_bs.txt 2> $O/bench_bs.err; tail -1 $O/bench_bs.txt | cut -c1-400
CLHIP_BS=0 timeout 300 $B > $O/bench_nobs.txt 2> $O/bench_nobs.err; tail -1 $O/bench_nobs.txt | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --deselect tests/test_gpu_bs.py > $O/test_all.txt 2>&1; echo "test_all rc $?"; tail -30 $O/test_all.txt
