#!/bin/bash
# Candidates prepared at the end of round 4 without GPU time left to measure them — run this first in the next GPU session.
#   CLHIP_WGRED_WIDE=1   slab reduction with 1 KB per wave and slab (csrc/conv3x3_wgrad.hip): other summation association (other bits,
#                        deterministic); parity subset below must stay green, then compare the bench step
set -u
mkdir -p gpurun_out/r05c; export TMPDIR=/tmp
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --no-sweep"
for v in default wgred_wide default; do
  if [ $v = wgred_wide ]; then export CLHIP_WGRED_WIDE=1; else unset CLHIP_WGRED_WIDE; fi
  echo "== $v"
  [ $v = wgred_wide ] && timeout 300 python -m pytest tests/test_gpu_wino.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "weight_gradient or golden_g1 or (full_size_vs_oracle and small)" 2>&1 | tail -2
  timeout 200 $B 2>/dev/null | cut -c1-200
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/r05c_$v -- python $OLDPWD/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-configs --no-sweep > /dev/null 2>&1; grep -h "wgrad_reduce" $PWD/r05c_$v/*/*kernel_stats.csv | cut -c1-160; rm -rf $PWD/r05c_$v )
done
