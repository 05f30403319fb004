#!/bin/bash
# round 6, second session: the CPU-vs-HIP chain (bench.chain_start / chain_collect) — its GPU test, then the pair + chain alone as the
# bench runs them (--sweep-only --sweep-tasks 0)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sweep_paths.py -x -q -k chain > gpurun_out/r06b_chain_test.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06b_chain_test.log
tail -5 gpurun_out/r06b_chain_test.log
timeout 1200 python bench.py --sweep-only --sweep-tasks 0 > gpurun_out/r06b_chain_sweep_only.json 2> gpurun_out/r06b_chain_sweep_only.err
echo "bench rc $?"
tail -c 6000 gpurun_out/r06b_chain_sweep_only.json
tail -5 gpurun_out/r06b_chain_sweep_only.err
