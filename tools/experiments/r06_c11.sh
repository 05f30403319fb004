#!/bin/bash
# round 6, third session, call 11: the driver's bench command with the pair / chain at a 4-epoch cap (6 before): seconds of the whole command, agreement figures
set -u
mkdir -p gpurun_out/r06c11; export TMPDIR=/tmp
SECONDS=0
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r06c11/bench.err > gpurun_out/r06c11/bench.json; echo "bench: $SECONDS s, $(wc -c < gpurun_out/r06c11/bench.json) bytes"
cp gpurun_out/bench_details.json gpurun_out/r06c11/bench_details.json
python - <<PY
import json
d = json.loads(open("gpurun_out/r06c11/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], json.dumps(d["sweep_s"]))
print(json.dumps(d["roofline"]["mfma_busy_pmc"]), d["roofline"]["frac"])
PY
