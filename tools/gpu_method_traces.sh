#!/bin/bash
# rocprofv3 kernel traces of one training batch of HAT / PackNet (wide_VGG9) and MAS / SI (base_VGG9).  usage: gpu_method_traces.sh <tag>
export TMPDIR=/tmp
TAG=${1:-r03}; P=$PWD; mkdir -p gpurun_out
for spec in "hat 64" "packnet 64" "mas 64" "si 64"; do
  set -- $spec
  d=$P/gpurun_out/${TAG}_$1_prof
  timeout 200 python tools/method_steps.py $1 $2 10 2>&1 | tail -1
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $P/tools/method_steps.py $1 $2 10 > /dev/null 2>&1 )
  f=$(find $d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_$1_kernel_stats.csv && python tools/prof_stats.py gpurun_out/${TAG}_$1_kernel_stats.csv 40
  rm -rf $d
done
