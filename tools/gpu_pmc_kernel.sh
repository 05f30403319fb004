#!/bin/bash
# SQ counters of ONE conv entry point (tools/one_kernel.py arguments): MFMA-pipe busy, wave cycles parked / stalled / issuing, VALU and
# VMEM issue, in two rocprofv3 --pmc passes (kernel trace only).   usage: gpu_pmc_kernel.sh <tag> <one_kernel.py args...>
set -u
TAG=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
P=$PWD
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  d=$P/gpurun_out/${TAG}_pmc_raw
  rm -rf $d
  ( cd /tmp && timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -- python $P/tools/one_kernel.py "$@" > $P/gpurun_out/${TAG}_pmc.log 2>&1 )
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set); dur = collections.defaultdict(float)
for r in rows:
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); k = re.sub(r"^void ", "", k); k = re.sub(r"\(.*", "", k)[:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in disp[k]:
        disp[k].add(r["Dispatch_Id"]); dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k in sorted(agg, key=lambda k: -dur[k]):
    if not re.search("conv|wgrad", k):
        continue
    n = len(disp[k])
    print("%s | dispatches %d | avg_us %.1f | %s" % (k, n, dur[k] / n / 1e3, "  ".join("%s=%.4g" % (c, v / n) for c, v in sorted(agg[k].items()))))
PY
  rm -rf $d
done
