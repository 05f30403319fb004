#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python bench.py --no-cpu-baseline --no-configs --no-sweep --steps 50 --warmup 10 2>gpurun_out/r03k.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); r=d['roofline']
print({k:v for k,v in r.items() if k not in ('per_layer','per_kernel')})
for x in r['per_layer']: print(x['kernel'], x['layer'], x['instance'][:50], round(x['us'],1), round(x['tflops'],1))"
tail -3 gpurun_out/r03k.err
bash tools/gpu_traffic.sh r03k_dgrad wino_dgrad_unpool 200 64 64 32 5
bash tools/gpu_traffic.sh r03k_wgrad wino_wgrad_unpool 200 64 64 32 5
bash tools/gpu_traffic.sh r03k_fwd wino_fwdpool 200 64 64 32 5
