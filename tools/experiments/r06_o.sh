#!/bin/bash
# round 6, session O: ablations of the bf16-split weight gradient on layer 2 (64 -> 64 @32x32, N = 200): no MFMAs / no splits / no loads /
# no splits + no loads / no sched_group_barrier / no SLP vectorisation
set -u
mkdir -p gpurun_out
for v in "" abl1 abl2 abl4 abl6 nosched noslp; do
  if [ -n "$v" ]; then export CLHIP_LIB=$PWD/clsurvey_amd/libclhip_$v.so; fi
  echo "== ${v:-product}"; timeout 300 python tools/bs_wgrad_bench.py 200 64 64 32 200 128 128 16 2>&1 | tail -2
done | tee gpurun_out/r06_o_abl.txt
