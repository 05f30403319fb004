#!/bin/bash
# session T: backward of the deep layers as ONE grid (wino_pair_kernel): parity of the operator, bit-identity of the plan executor's
# gradients against the build without merged grids (libclhip_nopair.so = -DCLHIP_PAIR_MAX_PIXELS=0), bench step A/B, kernel stats
set -u
mkdir -p gpurun_out/r05t; export TMPDIR=/tmp
O=gpurun_out/r05t; P=$PWD
timeout 300 python -m pytest tests/test_gpu_pair.py -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -15 | cut -c1-300
timeout 200 python tools/experiments/pair_engine_check.py dump /tmp/a.pt 2>&1 | grep -v amdgpu.ids | tail -2
CLHIP_LIB=$P/clsurvey_amd/libclhip_nopair.so timeout 200 python tools/experiments/pair_engine_check.py dump /tmp/b.pt 2>&1 | grep -v amdgpu.ids | tail -2
python tools/experiments/pair_engine_check.py cmp /tmp/a.pt /tmp/b.pt | tee $O/engine_gradients_pair_vs_nopair.txt
for r in 1 2; do
  for lib in libclhip.so libclhip_nopair.so; do
    CLHIP_LIB=$P/clsurvey_amd/$lib timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-sweep --no-configs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'ms_per_step %.4f' % d['ms_per_step'], 'value %.0f' % d['value'])"
  done
done | tee $O/bench_ab.txt
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$O/prof -- python $P/tools/one_step.py 20 small_VGG9_cl_128_128 > $P/$O/prof.log 2>&1 )
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_by_grid.py "$f" > $O/kernel_stats_by_grid.csv
rm -rf $O/prof
head -24 $O/kernel_stats_by_grid.csv | cut -c1-160
