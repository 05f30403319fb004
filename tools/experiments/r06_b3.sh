#!/bin/bash
# round 6, second session, call 3: first-layer weight gradient (conv3x3_wgrad_c3_unpool_kernel) — build before this session (u3old), the
# build with scalar-offset addressing + AGPR-resident accumulator (main; bitwise u3old), the same + bias sums on the matrix pipe (u3mfma);
# fc_tail rows per workgroup 32 / 16 / 8 by rocprofv3
set -u
mkdir -p gpurun_out/r06b3; export TMPDIR=/tmp
O=gpurun_out/r06b3; P=$PWD
for v in u3old main u3mfma; do
  L=clsurvey_amd/libclhip_$v.so; [ $v = main ] && L=clsurvey_amd/libclhip.so
  echo "== $v"
  CLHIP_LIB=$L python tools/experiments/u3_dump.py $O/dump_$v.npz 2>&1 | grep -v amdgpu.ids
  CLHIP_LIB=$L timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "wgrad or engine_matches or full_size" 2>&1 | tail -2
  CLHIP_LIB=$L timeout 120 python tools/conv_bench.py small 200 20 2>&1 | grep -E "3x64@64|^ALL"
  CLHIP_LIB=$L timeout 120 python tools/conv_bench.py small 200 20 2>&1 | grep -E "3x64@64|^ALL"
done
python - <<'PY'
import numpy as np
a=np.load("gpurun_out/r06b3/dump_u3old.npz"); b=np.load("gpurun_out/r06b3/dump_main.npz"); c=np.load("gpurun_out/r06b3/dump_u3mfma.npz")
print("main bitwise u3old:", all(np.array_equal(a[k], b[k]) for k in a.files))
print("u3mfma dw bitwise u3old:", all(np.array_equal(a[k], c[k]) for k in a.files if k.endswith("_dw")))
print("u3mfma db max rel diff:", max(float(np.abs(a[k]-c[k]).max()/max(np.abs(a[k]).max(),1e-30)) for k in a.files if k.endswith("_db")))
PY
for r in 32 16 8; do
  ( cd /tmp && CLHIP_FC_TAIL_ROWS=$r timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$O/prof$r -- python $P/tools/one_step.py 40 small_VGG9_cl_128_128 > $P/$O/prof$r.log 2>&1 )
  f=$(find $O/prof$r -name "*kernel_stats.csv" | head -1)
  python - "$f" $r <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "fc_tail" in r["Name"] or "fc_bwd_combo" in r["Name"] or "wgrad_c3" in r["Name"]:
        print("rows=%s %-40s calls %s avg %.2f us min %.2f" % (sys.argv[2], r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
tot=sum(float(r["TotalDurationNs"]) for r in rows if not r["Name"].startswith(("__amd","void at::")))
print("rows=%s kernel time per pass: %.1f us" % (sys.argv[2], tot/40/1e3))
PY
  rm -rf $O/prof$r
done
