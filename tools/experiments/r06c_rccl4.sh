#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
B="bench.py --gpus 1 --steps 20 --warmup 5 --no-sweep --no-configs --no-cpu-baseline"
p=29580
for skip in 0; do
for rep in 1 2; do
p=$((p+1))
echo -n "skip=$skip "
CLHIP_DBG_SKIP=$skip CLHIP_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $p $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], d['roofline']['avg_launch_us'])"
python -c "
import json
g = json.load(open('gpurun_out/bench_details.json'))['grid']
print(json.dumps(g['host_marks_s_rank0']))"
done
done
echo -n "plain "; python $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], d['roofline']['avg_launch_us'])"
