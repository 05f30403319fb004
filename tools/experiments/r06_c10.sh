#!/bin/bash
# round 6, third session, call 10: idle gaps between the launches of a pass (rocprofv3 kernel trace of 24 plan-executor passes; tools/trace_gaps.py)
set -u
mkdir -p gpurun_out/r06c10; export TMPDIR=/tmp
P=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $P/gpurun_out/r06c10/prof -- python $P/tools/one_step.py 24 small_VGG9_cl_128_128 > $P/gpurun_out/r06c10/prof.log 2>&1 )
f=$(find gpurun_out/r06c10/prof -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py "$f" | tee gpurun_out/r06c10_gaps.txt
rm -rf gpurun_out/r06c10/prof
