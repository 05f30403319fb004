#!/bin/bash
# MFMA utilisation per kernel of the engine step (final build): one rocprofv3 --pmc pass (counters only + kernel trace)
# usage: gpu_mfma_util.sh [model name] [output tag]
set -u
MODEL=${1:-small_VGG9_cl_128_128}
TAG=${2:-mfma_util}
export TMPDIR=/tmp
mkdir -p gpurun_out
P=$PWD
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $P/gpurun_out/mfma_util -- python $P/tools/one_step.py 4 $MODEL > $P/gpurun_out/mfma_util.log 2>&1 )
f=$(find gpurun_out/mfma_util -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY' | tee gpurun_out/${TAG}.csv
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
dur = collections.defaultdict(float)
for r in rows:
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); k = re.sub(r"^void ", "", k); k = re.sub(r"\(.*", "", k)[:72]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in disp[k]:
        disp[k].add(r["Dispatch_Id"]); dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
print("kernel,dispatches,avg_us,mfma_busy_over_32x_sq_busy,valu_active_over_wave_cycles")
for k in sorted(agg, key=lambda k: -dur[k]):
    a = agg[k]; n = len(disp[k])
    if a.get("SQ_BUSY_CYCLES", 0) <= 0 or not re.search("conv|wgrad|gemm|fc_chain|pair", k):
        continue
    # SQ_BUSY_CYCLES is summed over the 32 shader engines, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
    util = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (32.0 * a["SQ_BUSY_CYCLES"])
    valu = a["SQ_ACTIVE_INST_VALU"] / max(a["SQ_WAVE_CYCLES"], 1.0)
    print("%s,%d,%.1f,%.3f,%.3f" % (k, n, dur[k] / n / 1e3, util, valu))
PY
rm -rf gpurun_out/mfma_util
