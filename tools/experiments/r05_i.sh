#!/bin/bash
# Round 5, GPU session I: after the switch clean-up — whole GPU suite; HBM traffic + MFMA-busy PMC passes for the dominant launches
# (layer 2 of the bench net: Winograd weight gradient, bf16-split forward / backward-data).
set -u
mkdir -p gpurun_out/r05i; export TMPDIR=/tmp
O=gpurun_out/r05i
P=$PWD
timeout 1700 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/test_all.txt 2>&1; echo "test_all rc $?"; tail -6 $O/test_all.txt
for kind in wino_wgrad_unpool bs_fwdpool bs_dgrad_unpool; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $P/$O/t_${kind}_$ctr -- python $P/tools/one_kernel.py $kind 200 64 64 32 3 > /dev/null 2>&1 )
  done
  ( cd /tmp && timeout 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $P/$O/t_${kind}_SQ -- python $P/tools/one_kernel.py $kind 200 64 64 32 3 > /dev/null 2>&1 )
done
python - <<'PY' | tee gpurun_out/r05i/pmc_summary.txt
import csv, glob, collections, re
out = collections.OrderedDict()
for d in sorted(glob.glob("gpurun_out/r05i/t_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set); dur = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", re.sub(r"^void |\(anonymous namespace\)::", "", r["Kernel_Name"]))[:70]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in disp[k]:
                disp[k].add(r["Dispatch_Id"]); dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        for k, a in agg.items():
            if re.search("bs_conv|wino_wgrad", k):
                n = len(disp[k])
                print(d.split("/")[-1], "|", k, "| dispatches", n, "| avg_us %.1f |" % (dur[k] / n / 1e3), {c: "%.5g" % (v / n) for c, v in sorted(a.items())})
PY
rm -rf gpurun_out/r05i/t_*
