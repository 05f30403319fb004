#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/alexnet_step.py 128 10 2>&1 | tail -3 | tee gpurun_out/alex_step.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/alex_prof -- python $OLDPWD/tools/alexnet_step.py 128 5 > /dev/null 2> $OLDPWD/gpurun_out/alex_prof.err )
f=$(find gpurun_out/alex_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" | cut -c1-220
find gpurun_out/alex_prof -type f ! -name "*stats*" -size +1M -delete
