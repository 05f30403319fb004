#!/bin/bash
# Round 5, GPU session M: the 10-task sweep's OUTCOME (accepted lambdas, accuracies, forgetting) on three fp32-grade kernel paths —
# bf16-split convolutions nowhere / on the large maps (default) / everywhere; same task files, same seeds.
set -u
mkdir -p gpurun_out/r05m; export TMPDIR=/tmp
for m in 0 1 2; do
  CLHIP_BS=$m timeout 600 python bench.py --sweep-only --no-cpu-baseline > gpurun_out/r05m/sweep_bs$m.json 2> gpurun_out/r05m/sweep_bs$m.err
  python - <<PY
import json
s = json.loads(open("gpurun_out/r05m/sweep_bs$m.json").read().strip().splitlines()[-1])
print("CLHIP_BS=$m", round(s["gpu_s"], 1), "s", s["gpu_phase2_trainings_per_task"], s["gpu_accepted_lambda_per_task"], "avg acc %.1f forgetting %.1f" % (s["gpu_avg_accuracy"], s["gpu_avg_forgetting"]))
print("   final", [round(a, 1) for a in s["gpu_final_accuracies"]])
PY
done
