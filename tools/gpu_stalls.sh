#!/bin/bash
# Where do the waves of each kernel of the engine step spend their cycles?  One rocprofv3 --pmc pass (8 SQ counters, kernel
# trace only): parked at s_waitcnt / barriers (WAIT_ANY), issue stalls (WAIT_INST_ANY, of which LDS), active, VALU share,
# LDS bank-conflict cycles over LDS-active cycles.   usage: gpu_stalls.sh [model name] [output tag]
set -u
MODEL=${1:-small_VGG9_cl_128_128}
TAG=${2:-stalls}
export TMPDIR=/tmp
mkdir -p gpurun_out
P=$PWD
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $P/gpurun_out/stalls_raw -- python $P/tools/one_step.py 4 $MODEL > $P/gpurun_out/stalls.log 2>&1 )
f=$(find gpurun_out/stalls_raw -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY' | tee gpurun_out/${TAG}.csv
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
dur = collections.defaultdict(float)
for r in rows:
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); k = re.sub(r"^void ", "", k); k = re.sub(r"\(.*", "", k)[:72]
    k = k + " grid=" + r.get("Grid_Size", "?")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in disp[k]:
        disp[k].add(r["Dispatch_Id"]); dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
print("kernel|dispatches|avg_us|wait_any|wait_inst_any|wait_inst_lds|active_any|active_valu|lds_conflict_over_lds_active")
for k in sorted(agg, key=lambda k: -dur[k]):
    a = agg[k]; n = len(disp[k]); w = max(a.get("SQ_WAVE_CYCLES", 0), 1.0)
    if not re.search("conv|wgrad|gemm|fc_", k):
        continue
    print("%s|%d|%.1f|%.3f|%.3f|%.3f|%.3f|%.3f|%.3f" % (k, n, dur[k] / n / 1e3, a["SQ_WAIT_ANY"] / w, a["SQ_WAIT_INST_ANY"] / w, a["SQ_WAIT_INST_LDS"] / w,
          a["SQ_ACTIVE_INST_ANY"] / w, a["SQ_ACTIVE_INST_VALU"] / w, a["SQ_LDS_BANK_CONFLICT"] / max(a["SQ_LDS_IDX_ACTIVE"], 1.0)))
PY
rm -rf gpurun_out/stalls_raw
