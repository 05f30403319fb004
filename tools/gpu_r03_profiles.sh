#!/bin/bash
# round-3 profile set: default bench (what the driver runs), rocprofv3 kernel stats of the bench step, per-layer timings and
# MFMA-pipe busy (PMC) for the three VGG9 widths, kernel traces of one HAT / PackNet / MAS / SI batch, Winograd vs direct table.
set -u
export TMPDIR=/tmp
TAG=${1:-r03}; P=$PWD; mkdir -p gpurun_out/$TAG
( time timeout 900 python bench.py 2> gpurun_out/$TAG/bench.err > gpurun_out/$TAG/bench.json ) 2>&1 | tail -3
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/gpurun_out/$TAG/prof -- python $P/bench.py --no-cpu-baseline --no-configs --no-sweep > $P/gpurun_out/$TAG/prof_bench.json 2> $P/gpurun_out/$TAG/prof.err )
f=$(find gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$TAG/kernel_stats.csv
f=$(find gpurun_out/$TAG/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_by_grid.py "$f" > gpurun_out/$TAG/kernel_stats_by_grid.csv
rm -rf gpurun_out/$TAG/prof
for m in small base wide; do
  timeout 120 python tools/conv_bench.py $m 200 20 2>&1 | grep -v amdgpu.ids > gpurun_out/$TAG/conv_layers_$m.txt
  bash tools/gpu_mfma_util.sh ${m}_VGG9_cl_$([ $m = small ] && echo 128_128 || echo 512_512) $TAG/mfma_util_$m > /dev/null 2>&1
done
timeout 300 python tools/wino_bench.py 10 2>&1 | grep -v amdgpu.ids > gpurun_out/$TAG/wino_bench.txt
timeout 300 python tools/wino_bench.py 10 alex 2>&1 | grep -v amdgpu.ids >> gpurun_out/$TAG/wino_bench.txt
timeout 300 python tools/conv2d_bench.py 10 2>&1 | grep -v amdgpu.ids > gpurun_out/$TAG/conv2d_bench.txt
bash tools/gpu_alex.sh $TAG/alexnet 2>&1 | grep -v amdgpu.ids > gpurun_out/$TAG/alexnet_step.txt
for spec in "hat 64" "packnet 64" "mas 64" "si 64" "hat 224"; do
  set -- $spec
  d=$P/gpurun_out/$TAG/$1$2_prof
  timeout 200 python tools/method_steps.py $1 $2 10 2>&1 | tail -1 >> gpurun_out/$TAG/method_steps.txt
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $P/tools/method_steps.py $1 $2 8 > /dev/null 2>&1 )
  f=$(find $d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$TAG/$1$2_kernel_stats.csv
  rm -rf $d
done
python - <<PY
import json
d = json.loads(open("gpurun_out/$TAG/bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_us"], r["frac"], r["traffic"])
c = d.get("configs") or {}
for k, v in c.items():
    if isinstance(v, dict) and "ms_per_step" in v: print(k, round(v["ms_per_step"], 3), round(v["frac_of_f32_mfma_peak"], 3))
s = d.get("sweep") or {}
print({k: v for k, v in s.items() if k not in ("what", "pair")})
print(s.get("pair"))
PY
cat gpurun_out/$TAG/method_steps.txt
ls gpurun_out/$TAG
