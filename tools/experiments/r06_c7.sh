#!/bin/bash
# round 6, third session, call 7: sc1 output stores also in wino.hip (conv epilogues, weight-gradient slabs): parity, then the three widths A/B
# against libclhip_st0.so (all four sources with -DCLHIP_ST_AUX=0)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1700 python -m pytest tests/test_gpu_wino.py tests/test_gpu_pair.py tests/test_gpu_wide.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/r06c7_tests.log
for rep in 1 2; do
for lib in libclhip.so libclhip_st0.so; do
  echo "== $lib"
  for m in small base wide; do
    CLHIP_LIB=$PWD/clsurvey_amd/$lib timeout 300 python tools/conv_bench.py $m 200 20 2>&1 | grep -E "^ALL|TOTAL" | tr '\n' ' '; echo " ($m)"
  done
  CLHIP_LIB=$PWD/clsurvey_amd/$lib timeout 300 python bench.py --no-cpu-baseline --no-configs --no-sweep --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step ms', d['ms_per_step'], 'img/s', d['value'])"
done
done 2>&1 | tee gpurun_out/r06c7_ab.txt
