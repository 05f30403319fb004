#!/bin/bash
# round-3 closing run: the GPU test suite, smoke(), the default bench (what the driver runs) and its rocprofv3 kernel trace
set -u
export TMPDIR=/tmp
TAG=${1:-r03e}; P=$PWD; mkdir -p gpurun_out/$TAG
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python __graft_entry__.py --smoke 2>&1 | tail -2
( time timeout 900 python bench.py 2> gpurun_out/$TAG/bench.err > gpurun_out/$TAG/bench.json ) 2>&1 | tail -3
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/gpurun_out/$TAG/prof -- python $P/bench.py --no-cpu-baseline --no-configs --no-sweep > $P/gpurun_out/$TAG/prof_bench.json 2> $P/gpurun_out/$TAG/prof.err )
f=$(find gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/$TAG/kernel_stats.csv
f=$(find gpurun_out/$TAG/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_by_grid.py "$f" > gpurun_out/$TAG/kernel_stats_by_grid.csv
rm -rf gpurun_out/$TAG/prof
head -12 gpurun_out/$TAG/kernel_stats_by_grid.csv
python - <<PY
import json
d = json.loads(open("gpurun_out/$TAG/bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_us"], r["frac"], r["traffic"])
s = d.get("sweep") or {}
print({k: v for k, v in s.items() if k not in ("what", "pair")})
print({k: v for k, v in (s.get("pair") or {}).items() if k != "what"})
PY
