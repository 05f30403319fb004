#!/bin/bash
# round 6, second session, call 4: first-layer forward with the bias in the spare K slot + the cheaper pooling epilogue (main) against the
# build before this session (u3old); fc_tail at 16 rows per workgroup as the default; then the whole GPU suite on the new build
set -u
mkdir -p gpurun_out/r06b4; export TMPDIR=/tmp
O=gpurun_out/r06b4; P=$PWD
for v in u3old main u3old main; do
  L=clsurvey_amd/libclhip_$v.so; [ $v = main ] && L=clsurvey_amd/libclhip.so
  echo "== $v"
  CLHIP_LIB=$L timeout 120 python tools/conv_bench.py small 200 20 2>&1 | grep -E "3x64@64|^ALL"
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "fused_conv_relu_pool or engine_matches or full_size or g1" 2>&1 | tail -2
SECONDS=0
python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -x 2>&1 | tail -6 | cut -c1-300
echo "suite: $SECONDS s"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$O/prof -- python $P/tools/one_step.py 40 small_VGG9_cl_128_128 > $P/$O/prof.log 2>&1 )
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv; rm -rf $O/prof
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r06b4/kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows if not r["Name"].startswith(("__amd","void at::")))
for r in rows[:18]:
    print("%-70s calls %4s avg %8.2f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
print("kernel time per pass: %.1f us" % (tot/40/1e3))
PY
