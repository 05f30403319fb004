#!/bin/bash
# last GPU call of round 5 on HEAD: the driver's three commands (GPU suite, smoke, bench)
set -u
mkdir -p gpurun_out/r05zz; export TMPDIR=/tmp
O=gpurun_out/r05zz
SECONDS=0
python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3 | cut -c1-200
echo "suite: $SECONDS s"
python __graft_entry__.py --smoke 2>&1 | tail -1
SECONDS=0
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench.err > $O/bench.json; echo "bench: $SECONDS s, $(wc -c < $O/bench.json) bytes, $(wc -l < $O/bench.json) line(s)"
tail -1 $O/bench.json | cut -c1-700
