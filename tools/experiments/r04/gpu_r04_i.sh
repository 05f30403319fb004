#!/bin/bash
# round 4, call I: deeper unroll of the slab reduction — parity subset, bench line, kernel stats
set -u
mkdir -p gpurun_out/r04i; export TMPDIR=/tmp
O=gpurun_out/r04i; P=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wino.py -m gpu -q -x -p no:cacheprovider -k "engine or golden or wgrad or weight or determin or wino" 2>&1 | tail -3
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$O/prof -- python $P/bench.py --no-cpu-baseline --no-configs --no-sweep > $P/$O/prof_bench.json 2> $P/$O/prof.err )
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv
rm -rf $O/prof
grep -i "reduce\|fc_tail\|fc_bwd" $O/kernel_stats.csv | cut -c1-160
cut -c1-300 $O/prof_bench.json
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --no-sweep 2>/dev/null | cut -c1-260
