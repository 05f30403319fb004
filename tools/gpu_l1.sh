#!/bin/bash
mkdir -p gpurun_out/l1
for v in "" "$@"; do
  echo "== ${v:-base}"
  CLHIP_LIB=clsurvey_amd/libclhip${v:+_$v}.so timeout 120 python tools/conv_bench.py small 200 20 2>&1 | grep "3x64"
done
