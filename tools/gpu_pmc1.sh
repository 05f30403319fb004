#!/bin/bash
# PMC passes over ONE kernel: bash tools/gpu_pmc1.sh <tag> <one_kernel.py args...>
set -u
TAG=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
P=$PWD
pass() {  # name counters...
  n=$1; shift
  ( cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $P/gpurun_out/${TAG}_$n -- python $P/tools/${PMC_SCRIPT:-one_kernel.py} $ARGS > $P/gpurun_out/${TAG}_$n.log 2>&1 )
  f=$(find gpurun_out/${TAG}_$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in agg:
    n = len({r["Dispatch_Id"] for r in rows if r["Kernel_Name"][:60] == k})
    if "conv3x3" in k or "wgrad" in k or "fc_chain" in k:
        print(k, "dispatches", n, {c: round(v / n) for c, v in agg[k].items()})
PY
}
ARGS="$*"
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass b SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD
pass c SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_IFETCH SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
