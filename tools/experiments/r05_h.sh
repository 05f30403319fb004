#!/bin/bash
# Round 5, GPU session H: the adopted configuration (bf16-split convolutions on the large maps) — whole GPU suite, bench, profiles.
set -u
mkdir -p gpurun_out/r05h; export TMPDIR=/tmp
O=gpurun_out/r05h
P=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/test_all.txt 2>&1; echo "test_all rc $?"; tail -12 $O/test_all.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_stdout.txt 2> $O/bench_stderr.txt; echo "bench rc $? bytes $(wc -c < $O/bench_stdout.txt)"; tail -1 $O/bench_stdout.txt | cut -c1-1500
cp gpurun_out/bench_details.json $O/bench_details.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$O/prof -- python $P/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-configs --no-sweep > $P/$O/prof.log 2>&1 )
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv && head -25 $O/kernel_stats.csv | cut -c1-200
t=$(find $O/prof -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/trace_by_grid.py $t > $O/kernel_stats_by_grid.csv 2>/dev/null
rm -rf $O/prof
