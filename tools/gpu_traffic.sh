#!/bin/bash
# HBM traffic of ONE kernel by PMC (separate passes: FETCH_SIZE, WRITE_SIZE do not fit together; MI355X_MICROARCH.md
# "rocprofv3 PMC slots").  usage: bash tools/gpu_traffic.sh <tag> <one_kernel.py args...>
set -u
TAG=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
P=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $P/gpurun_out/${TAG}_$c -- python $P/tools/one_kernel.py "$@" > $P/gpurun_out/${TAG}_$c.log 2>&1 )
  f=$(find gpurun_out/${TAG}_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $c <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == sys.argv[2]]
agg = collections.defaultdict(list)
for r in rows:
    agg[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "conv3x3" in k or "wino" in k or "bs_" in k or "wgrad_reduce" in k:
        print(sys.argv[2], k, "dispatches", len(v), "per-dispatch", sorted(v)[len(v) // 2])
PY
done
