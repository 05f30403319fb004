#!/bin/bash
# round 4, call L: wave priority around the MFMA burst (CLHIP_W16G_PRIO = 1 ramp / 2 high during the burst / 3 high during staging)
set -u
mkdir -p gpurun_out/r04l; export TMPDIR=/tmp
for v in default prio1 prio2 prio3 default; do
  if [ $v = default ]; then unset CLHIP_LIB; else export CLHIP_LIB=$PWD/clsurvey_amd/libclhip_$v.so; fi
  for w in small wide; do
    echo "== conv_bench $w $v"; timeout 300 python tools/conv_bench.py $w 200 20 2>&1 | tail -22 > gpurun_out/r04l/conv_${w}_$v.txt; grep "64x64@32\|TOTAL\|ALL" gpurun_out/r04l/conv_${w}_$v.txt | cut -c1-150
  done
done
