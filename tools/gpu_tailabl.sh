#!/bin/bash
# phase costs of fc_tail_kernel: variants that return after phase k (garbage results, timing only)
export TMPDIR=/tmp
mkdir -p gpurun_out/tailabl
for v in "" "$@"; do
  d=$PWD/gpurun_out/tailabl/${v:-base}
  ( cd /tmp && CLHIP_LIB=$OLDPWD/clsurvey_amd/libclhip${v:+_$v}.so timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $OLDPWD/tools/one_step.py 20 > /dev/null 2>&1 )
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== ${v:-base}"; python - "$f" <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    if any(k in n for k in ("fc_tail", "fc_bwd_combo", "gemm_mfma", "reduce_multi", "fc_chain")):
        print("   %-50s %4s %8.1f us" % (n[:50], r["Calls"], float(r["AverageNs"]) / 1000))
PY
  rm -rf $d
done
