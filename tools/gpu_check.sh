#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof kernel stats. Outputs -> gpurun_out/
set -u
mkdir -p gpurun_out
TAG=${1:-r1}
export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -60 | tee gpurun_out/${TAG}_tests.log
echo "== smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -5 | tee gpurun_out/${TAG}_smoke.log
echo "== bench"; timeout 600 python bench.py --steps 100 --warmup 10 2> gpurun_out/${TAG}_bench.err | tee gpurun_out/${TAG}_bench.json
tail -5 gpurun_out/${TAG}_bench.err
echo "== rocprof"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/${TAG}_prof -- python $OLDPWD/bench.py --steps 30 --warmup 3 --no-cpu-baseline --kernel-iters 3 > $OLDPWD/gpurun_out/${TAG}_prof_bench.json 2> $OLDPWD/gpurun_out/${TAG}_prof.err )
find gpurun_out/${TAG}_prof -name "*kernel_stats*" | head; f=$(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
# keep the merged output small: drop the raw trace, keep stats
find gpurun_out/${TAG}_prof -type f ! -name "*stats*" -size +1M -delete
