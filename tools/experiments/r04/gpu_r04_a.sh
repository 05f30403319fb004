#!/bin/bash
# round 4, call A: staging-pipeline variants of wino_conv16g_kernel (CLHIP_W16G_PF = 0 / 1 / 2), the trajectory-separation
# test, and the 10-task sweep + pair on the 'blobs' task sequence
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04a
mkdir -p $O
echo "== wino + engine parity (default lib, PF=1)"
timeout 900 python -m pytest tests/test_gpu_wino.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "wino or engine or golden_g1 or fused_conv" 2>&1 | tail -5
echo "== same tests, PF=2 lib"
CLHIP_LIB=$PWD/clsurvey_amd/libclhip_pf2.so timeout 900 python -m pytest tests/test_gpu_wino.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for v in pf0 default pf2; do
  if [ $v = default ]; then unset CLHIP_LIB; else export CLHIP_LIB=$PWD/clsurvey_amd/libclhip_$v.so; fi
  echo "== conv_bench small $v"; timeout 300 python tools/conv_bench.py small 200 20 2>&1 | tail -22 > $O/conv_small_$v.txt; tail -4 $O/conv_small_$v.txt
  echo "== bench step $v"; timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-configs --no-sweep 2>/dev/null > $O/bench_$v.json; cut -c1-260 $O/bench_$v.json
done
for v in default pf2; do
  if [ $v = default ]; then unset CLHIP_LIB; else export CLHIP_LIB=$PWD/clsurvey_amd/libclhip_$v.so; fi
  echo "== conv_bench base/wide $v"; timeout 300 python tools/conv_bench.py base 200 10 2>&1 | tail -22 > $O/conv_base_$v.txt; tail -4 $O/conv_base_$v.txt
  timeout 300 python tools/conv_bench.py wide 200 10 2>&1 | tail -22 > $O/conv_wide_$v.txt; tail -4 $O/conv_wide_$v.txt
done
unset CLHIP_LIB
echo "== trajectory separation"
timeout 900 python -m pytest tests/test_gpu_trajectory.py -m gpu -q -x -s -p no:cacheprovider 2>&1 | tail -25 | tee $O/trajectory.txt
echo "== sweep (blobs) + pair"
timeout 1500 python bench.py --sweep-only 2> $O/sweep.err > $O/sweep.json; tail -3 $O/sweep.err; cut -c1-3000 $O/sweep.json
