#!/bin/bash
# whole GPU suite + smoke + the default bench run (what the driver runs at round end). Outputs -> gpurun_out/<tag>_*
set -u
TAG=${1:-full}; mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 ) 2>&1 | tee gpurun_out/${TAG}_tests.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3 | tee gpurun_out/${TAG}_smoke.log
( time timeout 900 python bench.py 2> gpurun_out/${TAG}_bench.err > gpurun_out/${TAG}_bench.json ) 2>&1 | tail -4
tail -2 gpurun_out/${TAG}_bench.err; python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], d.get("cpu_baseline"))
for k, v in (d.get("configs") or {}).items(): print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("ms_per_step", "images_per_s", "step_tflops", "step_frac_of_f32_mfma_peak")})
print(d.get("sweep_s"))
PY
