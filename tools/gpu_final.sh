#!/bin/bash
# the driver's round-end sequence + the rocprofv3 summary of the same bench command.  Outputs -> gpurun_out/<tag>_*
set -u
TAG=${1:-final}; mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python bench.py 2> gpurun_out/${TAG}_bench.err > gpurun_out/${TAG}_bench.json ) 2>&1 | tail -4
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/${TAG}_prof -- python $OLDPWD/bench.py --no-cpu-baseline --no-configs --no-sweep > $OLDPWD/gpurun_out/${TAG}_prof_bench.json 2> $OLDPWD/gpurun_out/${TAG}_prof.err )
f=$(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_kernel_stats.csv
find gpurun_out/${TAG}_prof -type f ! -name "*stats*" -size +1M -delete
python - <<PY
import json, csv
d = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_us"], r["back_to_back_us"], r["frac"], r["traffic"])
print(d.get("cpu_baseline", {}).get("value"), d["config"].get("gpu_over_cpu"), d.get("sweep_s"))
for k, v in (d.get("configs") or {}).items():
    if isinstance(v, dict) and "ms_per_step" in v: print(k, round(v["ms_per_step"], 3), round(v["images_per_s"]), round(v["frac_of_f32_mfma_peak"], 3))
for row in list(csv.DictReader(open("gpurun_out/${TAG}_kernel_stats.csv")))[:3]:
    print(row["Name"][:90], row["Calls"], float(row["AverageNs"]) / 1000)
PY
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
c = d.get("configs") or {}
for r in c.get("hbm_kernels", []): print(r["kernel"], round(r["us"], 1), "us", round(r["achieved_TBps"], 2), "TB/s", round(r["frac_of_hbm_peak"], 3))
for k, v in (c.get("conv_backward") or {}).items(): print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
s = d.get("sweep") or {}
print({k: v for k, v in s.items() if k not in ("what", "pair")})
print(s.get("pair"))
PY
