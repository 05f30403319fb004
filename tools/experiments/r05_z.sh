#!/bin/bash
# session Z: where the 29 us of fc_tail_kernel go: builds that return after stage 1..5 (CLHIP_TAIL_STOP; wrong results by design),
# duration of the launch by rocprofv3
set -u
mkdir -p gpurun_out/r05zz; export TMPDIR=/tmp
O=gpurun_out/r05zz; P=$PWD
for lib in libclhip.so libclhip_ts1.so libclhip_ts2.so libclhip_ts3.so libclhip_ts4.so libclhip_ts5.so; do
  ( cd /tmp && CLHIP_LIB=$P/clsurvey_amd/$lib timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$O/prof -- python $P/tools/one_step.py 10 small_VGG9_cl_128_128 > $P/$O/prof.log 2>&1 )
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
  python - "$f" "$lib" <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'fc_tail' in n or 'gemm_mfma' in n or 'fc_bwd_combo' in n:
        print(sys.argv[2], n.split('(')[1 if n.startswith('(') else 0][:40] if False else n[:60].replace('(anonymous namespace)::',''), 'calls', r['Calls'], 'avg_us %.1f' % (float(r['AverageNs'])/1e3))
PY
  rm -rf $O/prof
done 2>&1 | tee $O/fc_tail_stages.txt
