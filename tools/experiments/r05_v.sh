#!/bin/bash
# session V: per-launch durations of the merged backward grids under three block orders (rocprofv3 kernel trace of 20 executor steps)
set -u
mkdir -p gpurun_out/r05v; export TMPDIR=/tmp
O=gpurun_out/r05v; P=$PWD
for lib in libclhip.so libclhip_order1.so libclhip_order2.so libclhip_nopair.so; do
  ( cd /tmp && CLHIP_LIB=$P/clsurvey_amd/$lib timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$O/prof -- python $P/tools/one_step.py 20 small_VGG9_cl_128_128 > $P/$O/prof.log 2>&1 )
  f=$(find $O/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_by_grid.py "$f" > $O/by_grid_$lib.csv
  rm -rf $O/prof
  echo "== $lib"; grep "pair\|wino_wgrad_ps\|conv16" $O/by_grid_$lib.csv | cut -d, -f1-7 | cut -c1-150
done 2>&1 | tee $O/pair_orders.txt
