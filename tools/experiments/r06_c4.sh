#!/bin/bash
# round 6, third session, call 4: the 10-task sweep with NO CPU legs beside it (bench.py --sweep-only --no-cpu-baseline): seconds on an idle host
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py --sweep-only --no-cpu-baseline > gpurun_out/r06c4_sweep_alone.json 2> gpurun_out/r06c4_sweep_alone.err
python - <<PY | tee gpurun_out/r06c4_summary.txt
import json
d = json.loads([l for l in open("gpurun_out/r06c4_sweep_alone.json") if l.startswith("{")][-1])
print("alone gpu_s %.1f first_task %.1f passes %s acc %.1f forg %.2f" % (d.get("gpu_s", -1), d.get("gpu_first_task_s", -1), d.get("gpu_image_passes"), d.get("gpu_avg_accuracy", -1), d.get("gpu_avg_forgetting", -1)))
print({k: v for k, v in d.get("forced_paths", {}).items() if k != "per_task"} if isinstance(d.get("forced_paths"), dict) else None)
PY
