#!/bin/bash
# Round 5, GPU session D: where does the bf16-split conv kernel spend its time?  Timing-only ablations (BS_ABL variants built by
# tools/experiments/bs_ablate.py) on the layer-2 shape and an 8x8 shape, and one PMC pass (stall / LDS-conflict counters).
set -u
mkdir -p gpurun_out/r05d; export TMPDIR=/tmp
O=gpurun_out/r05d
P=$PWD
for shape in "200 64 64 32" "200 128 128 8"; do
  python tools/bs_layer.py $shape
  for a in 1 2 4 8 16 32 6 14 46 47; do
    CLHIP_LIB=$P/clsurvey_amd/libclhip_bsabl$a.so python tools/bs_layer.py $shape
  done
done 2>&1 | grep -v amdgpu.ids | tee $O/ablations.txt
for kind in bs_fwdpool bs_dgrad; do
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $P/$O/pmc_$kind -- python $P/tools/one_kernel.py $kind 200 64 64 32 3 > $P/$O/pmc_$kind.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $P/$O/pmc2_$kind -- python $P/tools/one_kernel.py $kind 200 64 64 32 3 > $P/$O/pmc2_$kind.log 2>&1 )
done
python - <<'PY' | tee gpurun_out/r05d/pmc_summary.txt
import csv, glob, collections, re
for d in sorted(glob.glob("gpurun_out/r05d/pmc*_bs_*")):
    if not d.endswith((".log",)):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
            for r in csv.DictReader(open(f)):
                k = re.sub(r"\(.*", "", re.sub(r"^void |\(anonymous namespace\)::", "", r["Kernel_Name"]))[:60]
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
            for k, a in agg.items():
                if "bs_conv" in k:
                    print(d.split("/")[-1], k, "dispatches", len(disp[k]), {c: "%.4g" % (v / len(disp[k])) for c, v in sorted(a.items())})
PY
rm -rf gpurun_out/r05d/pmc_bs_* gpurun_out/r05d/pmc2_bs_*
