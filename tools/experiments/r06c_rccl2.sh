#!/bin/bash
# round 6, third session: why is the step 2.85 ms in the one-rank RCCL dry run of bench.py (1.44 ms without a communicator)?
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { echo "== $1"; shift; "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], (d.get('grid') or {}).get('per_rank_ms_per_step'))"; }
B="bench.py --gpus 1 --steps 40 --warmup 5 --no-sweep --no-configs --no-cpu-baseline"
run "plain" python $B
run "torchrun + FORCE_DIST, 20 steps" env CLHIP_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29565 bench.py --gpus 1 --steps 20 --warmup 5 --no-sweep --no-configs --no-cpu-baseline
run "torchrun, no communicator" python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29562 $B
run "torchrun + FORCE_DIST" env CLHIP_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29563 $B
run "torchrun + FORCE_DIST + OMP_NUM_THREADS=8" env CLHIP_BENCH_FORCE_DIST=1 OMP_NUM_THREADS=8 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29564 $B
run "FORCE_DIST gloo" env CLHIP_BENCH_FORCE_DIST=1 CLHIP_BENCH_BACKEND=gloo python $B
