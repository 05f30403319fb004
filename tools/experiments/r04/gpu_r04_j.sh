#!/bin/bash
# round 4, call J: machine-scheduler strategy A/B (-mllvm -amdgpu-sched-strategy=max-ilp) on wino.hip (+ conv3x3_wgrad.hip)
set -u
mkdir -p gpurun_out/r04j; export TMPDIR=/tmp
O=gpurun_out/r04j
for v in default maxilp maxilp2; do
  if [ $v = default ]; then unset CLHIP_LIB; else export CLHIP_LIB=$PWD/clsurvey_amd/libclhip_$v.so; fi
  for w in small wide; do
    echo "== conv_bench $w $v"; timeout 300 python tools/conv_bench.py $w 200 20 2>&1 | tail -22 > $O/conv_${w}_$v.txt; tail -4 $O/conv_${w}_$v.txt
  done
  echo "== bench step $v"; timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --no-sweep 2>/dev/null > $O/bench_$v.json; cut -c1-240 $O/bench_$v.json
done
CLHIP_LIB=$PWD/clsurvey_amd/libclhip_maxilp2.so timeout 600 python -m pytest tests/test_gpu_wino.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "wino or engine or golden or wgrad" 2>&1 | tail -2
