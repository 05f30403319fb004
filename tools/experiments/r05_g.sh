#!/bin/bash
# Round 5, GPU session G: v5 of the bf16-split conv (weights straight from L2, in-wave staging) in three block / wave-tile shapes.
set -u
mkdir -p gpurun_out/r05g; export TMPDIR=/tmp
O=gpurun_out/r05g
P=$PWD
for cfg in 1 0 2; do
  export CLHIP_BS_CFG=$cfg
  timeout 900 python -m pytest tests/test_gpu_bs.py -m gpu -x -q -p no:cacheprovider > $O/test_bs_cfg$cfg.txt 2>&1; echo "cfg $cfg test_bs rc $?"; tail -3 $O/test_bs_cfg$cfg.txt
  timeout 300 python tools/bs_bench.py > $O/bs_bench_cfg$cfg.txt 2>&1; tail -16 $O/bs_bench_cfg$cfg.txt
done
export CLHIP_BS_CFG=1
for shape in "200 64 64 32"; do
  python tools/bs_layer.py $shape
  for a in 1 2 4 47; do
    CLHIP_LIB=$P/clsurvey_amd/libclhip_bsabl$a.so python tools/bs_layer.py $shape
  done
done 2>&1 | grep -v amdgpu.ids | tee $O/ablations_cfg1.txt
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-configs --no-sweep"
timeout 300 $B > $O/bench_bs.txt 2> $O/bench_bs.err; tail -1 $O/bench_bs.txt | cut -c1-300
