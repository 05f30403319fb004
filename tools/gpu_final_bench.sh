set -u
export TMPDIR=/tmp
TAG=r03g; P=$PWD; mkdir -p gpurun_out/$TAG
( time timeout 400 python bench.py 2> gpurun_out/$TAG/bench.err > gpurun_out/$TAG/bench.json ) 2>&1 | tail -3
( cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $P/gpurun_out/$TAG/prof -- python $P/bench.py --no-cpu-baseline --no-configs --no-sweep > $P/gpurun_out/$TAG/prof_bench.json 2> $P/gpurun_out/$TAG/prof.err )
f=$(find gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$TAG/kernel_stats.csv
f=$(find gpurun_out/$TAG/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_by_grid.py "$f" > gpurun_out/$TAG/kernel_stats_by_grid.csv
rm -rf gpurun_out/$TAG/prof
head -5 gpurun_out/$TAG/kernel_stats_by_grid.csv
python - <<PY
import json
d = json.loads(open("gpurun_out/$TAG/bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], d["config"]["step_frac_of_f32_mfma_peak"], r["kernel"], r["avg_launch_us"], r["frac"], r["traffic"])
s = d.get("sweep") or {}
print({k: v for k, v in s.items() if k in ("gpu_s","cpu_s_extrapolated","gpu_over_cpu_wall_clock","gpu_avg_forgetting")})
print({k: v for k, v in (s.get("pair") or {}).items() if k in ("gpu_s","cpu_s","gpu_accuracies","cpu_accuracies","gpu_over_cpu_wall_clock")})
c = d.get("configs") or {}
for k, v in c.items():
    if isinstance(v, dict) and "ms_per_step" in v: print(k, round(v["ms_per_step"], 3), round(v["frac_of_f32_mfma_peak"], 3))
print(d["cpu_baseline"]["value"], d["config"].get("gpu_over_cpu"))
PY
