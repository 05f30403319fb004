#!/bin/bash
set -u
mkdir -p gpurun_out/r05r; export TMPDIR=/tmp
P=$PWD
export CLHIP_LIB=$P/clsurvey_amd/libclhip_bsr16.so
timeout 600 python bench.py --sweep-only --no-cpu-baseline > gpurun_out/r05r/sweep.json 2> gpurun_out/r05r/sweep.err
python - <<'PY'
import json
s = json.loads(open("gpurun_out/r05r/sweep.json").read().strip().splitlines()[-1])
print("rule16", round(s["gpu_s"], 1), "s", s["gpu_phase2_trainings_per_task"], s["gpu_accepted_lambda_per_task"], "avg acc %.1f forgetting %.1f" % (s["gpu_avg_accuracy"], s["gpu_avg_forgetting"]))
print("   final", [round(a, 1) for a in s["gpu_final_accuracies"]])
PY
timeout 200 python tools/conv_bench.py small 200 20 2>&1 | grep -v amdgpu.ids | grep "@16\|ALL" | cut -c1-42,100-118
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --no-sweep 2>/dev/null | cut -c1-200
unset CLHIP_LIB
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --no-sweep 2>/dev/null | cut -c1-200
