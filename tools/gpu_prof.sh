#!/bin/bash
# bench + rocprofv3 kernel stats of the same command. Outputs -> gpurun_out/<tag>_*
set -u
mkdir -p gpurun_out
TAG=${1:-r2}
export TMPDIR=/tmp
timeout 600 python bench.py --steps 100 --warmup 10 ${2:-} 2> gpurun_out/${TAG}_bench.err > gpurun_out/${TAG}_bench.json
tail -3 gpurun_out/${TAG}_bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/${TAG}_prof -- python $OLDPWD/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-configs --no-sweep --kernel-iters 3 > $OLDPWD/gpurun_out/${TAG}_prof_bench.json 2> $OLDPWD/gpurun_out/${TAG}_prof.err )
f=$(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_kernel_stats.csv && head -30 "$f" | cut -c1-200
find gpurun_out/${TAG}_prof -type f ! -name "*stats*" -size +1M -delete
