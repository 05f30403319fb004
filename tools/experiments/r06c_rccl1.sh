#!/bin/bash
# round 6, third session: the N > 1 code paths over the REAL backend (RCCL) with a communicator of one rank — what a 1-GPU box can
# check of them: (1) tests/test_shard_gpu.py::test_driver_shard_one_rank_over_rccl (every collective of framework/shard.py forced at
# world 1, then the sharded driver), (2) bench.py's N > 1 branch under torch.distributed.run with one process (CLHIP_BENCH_FORCE_DIST=1)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_shard_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/r06c_rccl1_tests.log
SECONDS=0
CLHIP_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r06c_rccl1_bench.err > gpurun_out/r06c_rccl1_bench.json
echo "bench (one-rank RCCL dry run): $SECONDS s rc=$?"
tail -1 gpurun_out/r06c_rccl1_bench.json | cut -c1-2500
tail -5 gpurun_out/r06c_rccl1_bench.err | cut -c1-400
