#!/bin/bash
# round 6, third session, call 3: the bench's own process pinned to free cores while the CPU legs of the pair / chain run beside the
# GPU sweep (CLHIP_BENCH_PIN_SELF=1, the default) against unpinned (=0): `bench.py --sweep-only`, sweep seconds on the GPU
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for pin in 1 0; do
  CLHIP_BENCH_PIN_SELF=$pin timeout 900 python bench.py --sweep-only > gpurun_out/r06c3_sweep_pin$pin.json 2> gpurun_out/r06c3_sweep_pin$pin.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r06c3_sweep_pin$pin.json") if l.startswith("{")][-1])
print("pin=$pin gpu_s %.1f first_task %.1f pair gpu %.2f cpu %.1f pinned %s acc %.1f forg %.2f" % (d.get("gpu_s", -1), d.get("gpu_first_task_s", -1),
      d["pair"]["gpu_s"], d["pair"]["cpu_s"], d.get("gpu_process_pinned_to_logical_cpus"), d.get("gpu_avg_accuracy", -1), d.get("gpu_avg_forgetting", -1)))
PY
done | tee gpurun_out/r06c3_summary.txt
