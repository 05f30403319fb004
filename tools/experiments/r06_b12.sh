#!/bin/bash
# round 6, second session, call 12: the build with both bit-changing forms off (main) against the library of the session's start
# (r06a): whole-engine losses / logits / gradient arenas bitwise; first-layer timings; then the bench's sweep (pair + chain + 10 tasks)
set -u
mkdir -p gpurun_out/r06b12
for v in r06a main; do
  L=clsurvey_amd/libclhip_$v.so; [ $v = main ] && L=clsurvey_amd/libclhip.so
  CLHIP_LIB=$L python tools/experiments/engine_dump.py gpurun_out/r06b12/eng_$v.npz 2>&1 | grep -v amdgpu.ids | tail -1
  echo -n "$v: "; CLHIP_LIB=$L timeout 120 python tools/conv_bench.py small 200 20 2>&1 | grep -E "3x64@64|^ALL" | cut -c1-12,95-125 | tr '\n' ' '; echo
done
python - <<'PY'
import numpy as np
a=np.load("gpurun_out/r06b12/eng_r06a.npz"); b=np.load("gpurun_out/r06b12/eng_main.npz")
bad=[k for k in a.files if not np.array_equal(a[k], b[k])]
print("engine outputs bitwise the session-start library:", not bad, bad[:6])
PY
rm -f gpurun_out/r06b12/eng_*.npz
SECONDS=0
timeout 1500 python bench.py --sweep-only > gpurun_out/r06b12/sweep_only.json 2> gpurun_out/r06b12/sweep_only.err
echo "sweep-only: $SECONDS s rc $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06b12/sweep_only.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("gpu_s","gpu_avg_accuracy","gpu_avg_forgetting","gpu_error")})
print(d.get("conditioning")); print(d.get("gpu_accepted_lambda_per_task"))
fp=d.get("forced_paths",{}); print("forced:", fp.get("max_gap_all_tasks_points"), fp.get("well_conditioned"), fp.get("error"))
for e in fp.get("per_task",[]): print("  ", e["task"], e["lambda"], e.get("lambda_accepted_by_free_run"), round(e["x"],2), e["gap_points"], e["omega_sum_rel_spread"])
c=d.get("chain",{}); print("chain:", c.get("max_gap_points"), c.get("max_omega_sum_rel_gap"), c.get("error"))
p=d.get("pair",{}); print("pair:", p.get("gpu_s"), p.get("cpu_s"), p.get("max_accuracy_gap_points"))
PY
