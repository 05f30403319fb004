#!/bin/bash
# round 4, call D: A/B of wino_conv16g schedules (default PF=2 / MFMA burst + PF=1 / no SLP packing) per width
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04d
mkdir -p $O
for v in default burst1 noslp; do
  if [ $v = default ]; then unset CLHIP_LIB; else export CLHIP_LIB=$PWD/clsurvey_amd/libclhip_$v.so; fi
  for w in small wide; do
    echo "== conv_bench $w $v"; timeout 300 python tools/conv_bench.py $w 200 20 2>&1 | tail -22 > $O/conv_${w}_$v.txt; tail -4 $O/conv_${w}_$v.txt
  done
  echo "== bench step $v"; timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-configs --no-sweep 2>/dev/null > $O/bench_$v.json; cut -c1-260 $O/bench_$v.json
done
unset CLHIP_LIB
echo "== re-run of the three revised tests"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide.py -m gpu -q -p no:cacheprovider --tb=short -k "g34 or full_size or hat_step_wide" 2>&1 | tail -8 | cut -c1-300
echo "== default bench"
SECONDS=0
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench wall-clock: $SECONDS s"; tail -3 $O/bench.err | cut -c1-300; cut -c1-1500 $O/bench.json
