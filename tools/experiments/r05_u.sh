#!/bin/bash
# session U: merged backward grids, second build (8x8 layers with few in-channels take one-image blocks for the backward-data half):
# operator parity, bit-identity of the executor's gradients against the build without merged grids, bench step of the build and of
# three measurement builds (block order: one list after the other, either way round; 512 weight-gradient blocks on 8x8 maps),
# kernel stats, then the whole GPU suite
set -u
mkdir -p gpurun_out/r05u; export TMPDIR=/tmp
O=gpurun_out/r05u; P=$PWD
timeout 300 python -m pytest tests/test_gpu_pair.py -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -5 | cut -c1-300
timeout 200 python tools/experiments/pair_engine_check.py dump /tmp/a.pt 2>&1 | grep -v amdgpu.ids | tail -1
CLHIP_LIB=$P/clsurvey_amd/libclhip_nopair.so timeout 200 python tools/experiments/pair_engine_check.py dump /tmp/b.pt 2>&1 | grep -v amdgpu.ids | tail -1
python tools/experiments/pair_engine_check.py cmp /tmp/a.pt /tmp/b.pt | tee $O/engine_gradients_pair_vs_nopair.txt
for r in 1 2; do
  for lib in libclhip.so libclhip_nopair.so libclhip_order1.so libclhip_order2.so libclhip_t512.so; do
    CLHIP_LIB=$P/clsurvey_amd/$lib timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-sweep --no-configs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'ms_per_step %.4f' % d['ms_per_step'], 'value %.0f' % d['value'])"
  done
done | tee $O/bench_ab.txt
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$O/prof -- python $P/tools/one_step.py 20 small_VGG9_cl_128_128 > $P/$O/prof.log 2>&1 )
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_by_grid.py "$f" > $O/kernel_stats_by_grid.csv
rm -rf $O/prof
grep pair $O/kernel_stats_by_grid.csv | cut -c1-160
SECONDS=0
python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -x 2>&1 | tail -6 | cut -c1-300
echo "suite: $SECONDS s"
