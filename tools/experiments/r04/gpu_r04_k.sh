#!/bin/bash
# round 4, call K: TIMING-ONLY ablations of wino_conv16g_kernel's loop (CLHIP_W16G_ABL bit mask; results are wrong by design)
set -u
mkdir -p gpurun_out/r04k; export TMPDIR=/tmp
rm -f gpurun_out/r04k/ablations.txt
for v in default abl_tf abl_a abl_bar abl_st abl_ld abl_mem abl_all abl_mfma default; do
  if [ $v = default ]; then unset CLHIP_LIB; else export CLHIP_LIB=$PWD/clsurvey_amd/libclhip_$v.so; fi
  timeout 200 python tools/conv_bench.py small 200 20 2>&1 | grep "wino_conv16g" | python -c "
import re, sys
for ln in sys.stdin:
    m = re.search(r'^(\S+)\s+(\S+)\s+.*?([0-9.]+) us', ln)
    print('%-9s %-26s %-10s %7s us' % ('$v', m.group(1), m.group(2), m.group(3)))
" | tee -a gpurun_out/r04k/ablations.txt
done
