#!/bin/bash
# round-4 profile set: GPU test suite + smoke, default bench (what the driver runs) and the rocprofv3 kernel stats of the bench
# step, per-layer timings and MFMA-pipe busy (PMC) for the three VGG9 widths, stall counters, HBM traffic of the dominant launch,
# AlexNet step split.
set -u
export TMPDIR=/tmp
TAG=${1:-r04}; P=$PWD; mkdir -p gpurun_out/$TAG
SECONDS=0
python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -12 | cut -c1-300
echo "test suite: $SECONDS s"
python __graft_entry__.py --smoke 2>&1 | tail -2
SECONDS=0
timeout 900 python bench.py 2> gpurun_out/$TAG/bench.err > gpurun_out/$TAG/bench.json; echo "bench: $SECONDS s"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/gpurun_out/$TAG/prof -- python $P/bench.py --no-cpu-baseline --no-configs --no-sweep > $P/gpurun_out/$TAG/prof_bench.json 2> $P/gpurun_out/$TAG/prof.err )
f=$(find gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$TAG/kernel_stats.csv
f=$(find gpurun_out/$TAG/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_by_grid.py "$f" > gpurun_out/$TAG/kernel_stats_by_grid.csv
rm -rf gpurun_out/$TAG/prof
for m in small base wide; do
  timeout 120 python tools/conv_bench.py $m 200 20 2>&1 | grep -v amdgpu.ids > gpurun_out/$TAG/conv_layers_$m.txt
  bash tools/gpu_mfma_util.sh ${m}_VGG9_cl_$([ $m = small ] && echo 128_128 || echo 512_512) $TAG/mfma_util_$m > /dev/null 2>&1
done
bash tools/gpu_stalls.sh small_VGG9_cl_128_128 $TAG/stalls_small > /dev/null 2>&1
bash tools/gpu_alex.sh $TAG/alexnet 2>&1 | grep -v amdgpu.ids > gpurun_out/$TAG/alexnet_step.txt
head -14 gpurun_out/$TAG/kernel_stats_by_grid.csv | cut -c1-200
python - <<PY
import json
d = json.loads(open("gpurun_out/$TAG/bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_us"], r["frac"], r["traffic"])
s = d.get("sweep") or {}
print({k: v for k, v in s.items() if k not in ("what", "pair", "gpu_phase2_attempts")})
print({k: v for k, v in (s.get("pair") or {}).items() if k not in ("what", "gpu", "cpu", "cpu_other_threads")})
PY
ls gpurun_out/$TAG
