#!/bin/bash
# round 6, third session, call 8: full_sweep with the GPU sweep FIRST (idle host), then the pair / chain / kernel-path legs: `bench.py --sweep-only`
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
SECONDS=0; timeout 900 python bench.py --sweep-only > gpurun_out/r06c8_sweep.json 2> gpurun_out/r06c8_sweep.err
echo "wall $SECONDS s"; tail -2 gpurun_out/r06c8_sweep.err
python - <<PY | tee gpurun_out/r06c8_summary.txt
import json
d = json.loads([l for l in open("gpurun_out/r06c8_sweep.json") if l.startswith("{")][-1])
print("gpu_s %.1f first_task %.1f pair gpu %.2f cpu %.1f gap %.2f acc %.1f forg %.2f" % (d.get("gpu_s", -1), d.get("gpu_first_task_s", -1),
      d["pair"]["gpu_s"], d["pair"]["cpu_s"], d["pair"]["max_accuracy_gap_points"], d.get("gpu_avg_accuracy", -1), d.get("gpu_avg_forgetting", -1)))
print(d["pair"]["cpu_concurrency"])
ch = d.get("chain", {})
print("chain", {k: ch.get(k) for k in ("tasks_compared", "max_gap_points", "max_omega_sum_rel_gap", "cpu_over_gpu_seconds", "error")})
fp = d.get("forced_paths", {})
print("forced", {k: v for k, v in fp.items() if k != "per_task"} if isinstance(fp, dict) else fp)
print("error", d.get("error"), d.get("gpu_error"))
PY
