#!/bin/bash
# Round 5, GPU session F: persistent in-wave pipelined bf16-split conv: parity, per-layer
# timing, ablations on the layer-2 shape, step time.
set -u
mkdir -p gpurun_out/r05f; export TMPDIR=/tmp
O=gpurun_out/r05f
P=$PWD
timeout 900 python -m pytest tests/test_gpu_bs.py -m gpu -x -q -p no:cacheprovider > $O/test_bs.txt 2>&1; echo "test_bs rc $?"; tail -5 $O/test_bs.txt
timeout 300 python tools/bs_bench.py > $O/bs_bench.txt 2>&1; tail -16 $O/bs_bench.txt
for shape in "200 64 64 32" "200 64 64 16" "200 128 128 8"; do
  python tools/bs_layer.py $shape
  for a in 1 2 4 47; do
    CLHIP_LIB=$P/clsurvey_amd/libclhip_bsabl$a.so python tools/bs_layer.py $shape
  done
done 2>&1 | grep -v amdgpu.ids | tee $O/ablations.txt
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-configs --no-sweep"
timeout 300 $B > $O/bench_bs.txt 2> $O/bench_bs.err; tail -1 $O/bench_bs.txt | cut -c1-300
