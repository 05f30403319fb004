#!/bin/bash
# round 4, call M: fenced MFMA burst on the non-UNPOOL instances of wino_conv16g_kernel (the shipped default) vs CLHIP_W16G_PRIO=0
set -u
mkdir -p gpurun_out/r04m; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_wino.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "wino or engine or golden or full_size" 2>&1 | tail -2
for v in default prio0 default; do
  if [ $v = default ]; then unset CLHIP_LIB; else export CLHIP_LIB=$PWD/clsurvey_amd/libclhip_$v.so; fi
  for w in small base wide; do
    echo "== conv_bench $w $v"; timeout 300 python tools/conv_bench.py $w 200 20 2>&1 | tail -22 > gpurun_out/r04m/conv_${w}_$v.txt; grep "ALL" gpurun_out/r04m/conv_${w}_$v.txt
  done
  echo "== bench step $v"; timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --no-sweep 2>/dev/null | cut -c1-240
done
