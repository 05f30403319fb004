#!/bin/bash
# round-4 closing run: GPU suite, smoke, default bench, and a 2-rank dry run of the N > 1 bench path (gloo, both ranks on cuda:0)
set -u
mkdir -p gpurun_out/r04h; export TMPDIR=/tmp
O=gpurun_out/r04h
SECONDS=0
python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -4 | cut -c1-300
echo "suite: $SECONDS s"
python __graft_entry__.py --smoke 2>&1 | tail -1
SECONDS=0
timeout 900 python bench.py 2> $O/bench.err > $O/bench.json; echo "bench: $SECONDS s"
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["avg_launch_us"], r["frac"], r["mfma_issued_frac"], r["mfma_busy_pmc"])
s = d.get("sweep") or {}
p = s.get("pair") or {}
print(s.get("gpu_s"), s.get("gpu_phase2_trainings_per_task"), s.get("gpu_avg_accuracy"), s.get("gpu_avg_forgetting"), s.get("error"))
print({k: v for k, v in p.items() if k not in ("what", "gpu", "cpu", "cpu_other_threads")}, (p.get("cpu") or {}).get("pinned_to_logical_cpus"), (p.get("cpu_other_threads") or {}).get("pinned_to_logical_cpus"))
print(d["sharded_sweep"].get("seconds"), d["sharded_sweep"].get("error"))
PY
SECONDS=0
CLHIP_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --steps 20 --warmup 5 2> $O/bench2.err > $O/bench2.json; echo "bench --gpus 2 (gloo dry run): $SECONDS s"
tail -2 $O/bench2.err | cut -c1-300
python - <<PY
import json
txt = open("$O/bench2.json").read().strip().splitlines()
d = json.loads([l for l in txt if l.startswith("{")][-1])
print(d["n_gpus"], d["value"], d["ms_per_step"], d.get("grid", {}).get("fill_factor"))
print(json.dumps(d["sharded_sweep"])[:1500])
PY
