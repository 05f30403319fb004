#!/bin/bash
# Round 5, GPU session Q: ONE half-size task of the EWC sweep measured on both sides (tools/cpu_full_task.py; the quarter-size task of
# round 4 took 462 s on the host cores, the full-size one does not fit a GPU session of this pool)
set -u
mkdir -p gpurun_out/r05q; export TMPDIR=/tmp
CLHIP_CPUTASK_SIZES=4000,1000,500 timeout 1500 python tools/cpu_full_task.py --ranks 5 --threads 16 --out gpurun_out/r05q/cpu_half_task.json 2>&1 | grep -v amdgpu.ids | tail -5
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05q/cpu_half_task.json"))
print({k: v for k, v in d.items() if not isinstance(v, (list, dict))})
PY
