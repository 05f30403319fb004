#!/bin/bash
# round-4 closing run on the final build: GPU suite, smoke, default bench, 2-rank dry run of the N > 1 bench path (gloo, both ranks on
# cuda:0); then the profile set of the same build: rocprofv3 kernel trace of the bench step, MFMA-pipe busy (PMC) of small_VGG9's kernels,
# HBM traffic (PMC, separate passes) of the three layer-2 launches
set -u
mkdir -p gpurun_out/r04z; export TMPDIR=/tmp
O=gpurun_out/r04z; P=$PWD
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
print(d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_us"], r["frac"], r["mfma_issued_frac"], r["mfma_busy_pmc"], r["traffic"])
s = d.get("sweep") or {}
p = s.get("pair") or {}
print(s.get("gpu_s"), s.get("gpu_phase2_trainings_per_task"), s.get("gpu_avg_accuracy"), s.get("gpu_avg_forgetting"), s.get("error"))
print({k: v for k, v in p.items() if k not in ("what", "gpu", "cpu", "cpu_other_threads")})
print(d["sharded_sweep"].get("seconds"), d["sharded_sweep"].get("error"))
PY
SECONDS=0
CLHIP_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --steps 20 --warmup 5 2> $O/bench2.err > $O/bench2.json; echo "bench --gpus 2 (gloo dry run): $SECONDS s"
python - <<PY
import json
txt = open("$O/bench2.json").read().strip().splitlines()
d = json.loads([l for l in txt if l.startswith("{")][-1])
print(d["n_gpus"], d["value"], d["ms_per_step"], d["sharded_sweep"].get("seconds"), d["sharded_sweep"].get("error"))
PY
SECONDS=0
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$O/prof -- python $P/bench.py --no-cpu-baseline --no-configs --no-sweep > $P/$O/prof_bench.json 2> $P/$O/prof.err )
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_by_grid.py "$f" > $O/kernel_stats_by_grid.csv
rm -rf $O/prof
head -8 $O/kernel_stats_by_grid.csv | cut -c1-220
echo "trace: $SECONDS s"; SECONDS=0
bash tools/gpu_mfma_util.sh small_VGG9_cl_128_128 r04z/mfma_util_small > /dev/null 2>&1; head -6 $O/mfma_util_small.csv | cut -c1-150
echo "mfma util: $SECONDS s"; SECONDS=0
for k in wino_fwdpool wino_dgrad_unpool wino_wgrad_unpool; do
  bash tools/gpu_traffic.sh r04z/traffic_$k $k 200 64 64 32 5 2>&1 | grep -v amdgpu.ids | tee -a $O/traffic.txt | cut -c1-200
  rm -rf $O/traffic_${k}_FETCH_SIZE $O/traffic_${k}_WRITE_SIZE
done
echo "traffic: $SECONDS s"
