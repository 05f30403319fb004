#!/bin/bash
# AlexNet 224x224 train step under rocprofv3: per-kernel split.  usage: gpu_alex.sh <tag> [variant]
export TMPDIR=/tmp
TAG=${1:-alex}; V=${2:-}
d=$PWD/gpurun_out/${TAG}_prof
[ -n "$V" ] && export CLHIP_LIB=$PWD/clsurvey_amd/libclhip_$V.so
timeout 300 python tools/alexnet_step.py 128 10 2>&1 | tail -1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $OLDPWD/tools/alexnet_step.py 128 5 > /dev/null 2>&1 )
f=$(find $d -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/${TAG}_kernel_stats.csv
python - "$f" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:18]:
    n = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    print("%-72s %4s %8.1f us %5.1f%%" % (n[:72], r["Calls"], float(r["AverageNs"]) / 1000, 100 * float(r["TotalDurationNs"]) / tot))
PY
rm -rf $d
