#!/bin/bash
# round 6, third session, call 15: the library at the head of the round against the library at the start of the third session (commit 9f00b91,
# built as libclhip_s3start.so), alternating on one box: bench step (200 steps), rocprofv3 kernel time of a pass
set -u
mkdir -p gpurun_out/r06c15; export TMPDIR=/tmp
B="bench.py --no-cpu-baseline --no-configs --no-sweep --steps 200 --warmup 20"
for rep in 1 2 3; do
for lib in libclhip.so libclhip_s3start.so; do
  echo -n "$lib  "
  CLHIP_LIB=$PWD/clsurvey_amd/$lib timeout 300 python $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step ms %.4f' % d['ms_per_step'])"
done
done | tee gpurun_out/r06c15_ab.txt
P=$PWD
for lib in libclhip.so libclhip_s3start.so; do
( cd /tmp && CLHIP_LIB=$P/clsurvey_amd/$lib timeout 300 rocprofv3 --kernel-trace --output-format csv -d $P/gpurun_out/r06c15/prof_$lib -- python $P/tools/one_step.py 24 small_VGG9_cl_128_128 > $P/gpurun_out/r06c15/prof.log 2>&1 )
f=$(find gpurun_out/r06c15/prof_$lib -name "*kernel_trace.csv" | head -1)
echo "$lib: $(python tools/trace_gaps.py "$f" | head -1)" | tee -a gpurun_out/r06c15_ab.txt
rm -rf gpurun_out/r06c15/prof_$lib
done
