#!/bin/bash
# round 6, session S: schedule variants of bs_wgrad_kernel (VALU per MFMA of the sched_group_barrier pattern, none, no SLP packing)
set -u
mkdir -p gpurun_out
for v in "" vpm2 vpm4 vpm6 nosched noslp; do
  if [ -n "$v" ]; then export CLHIP_LIB=$PWD/clsurvey_amd/libclhip_$v.so; fi
  echo "== ${v:-product (3 VALU per MFMA)}"; timeout 300 python tools/bs_wgrad_bench.py 200 64 64 32 200 256 256 16 2>&1 | tail -2
done | tee gpurun_out/r06_s_sched.txt
