#!/bin/bash
# Round 5, GPU session A: (1) bf16-split micro-benchmark (numerics + issue rates), (2) the driver's bench command — is the LAST
# stdout line the compact record?, (3) round 4's unmeasured candidate CLHIP_WGRED_WIDE.
set -u
mkdir -p gpurun_out/r05a; export TMPDIR=/tmp
O=gpurun_out/r05a
hipcc --offload-arch=gfx950 -O3 -w tools/micro/bf16_split_dot.hip -o /tmp/bf16_split_dot && timeout 300 /tmp/bf16_split_dot > $O/bf16_split_dot.txt 2>&1
echo "== micro done"; tail -30 $O/bf16_split_dot.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_stdout.txt 2> $O/bench_stderr.txt
echo "bench rc $? stdout bytes $(wc -c < $O/bench_stdout.txt) lines $(wc -l < $O/bench_stdout.txt)"
tail -1 $O/bench_stdout.txt | cut -c1-600
cp gpurun_out/bench_details.json $O/ 2>/dev/null
bash tools/gpu_r05_candidates.sh > $O/candidates.txt 2>&1
tail -20 $O/candidates.txt
