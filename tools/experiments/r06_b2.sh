#!/bin/bash
# round 6, second session, call 2: chain test at the bench's size; fc_tail rows-per-workgroup A/B (bitwise tests + rocprofv3 averages);
# pair + chain as the bench runs them with the chain's CPU legs on the second socket
set -u
mkdir -p gpurun_out/r06b2; export TMPDIR=/tmp
O=gpurun_out/r06b2; P=$PWD
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_sweep_paths.py -x -q -k chain 2>&1 | tail -3
echo "chain test: $SECONDS s"
for r in 32 16 8; do
  CLHIP_FC_TAIL_ROWS=$r timeout 600 python -m pytest tests/test_gpu_fc_tail.py -x -q 2>&1 | tail -1
  ( cd /tmp && CLHIP_FC_TAIL_ROWS=$r timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$O/prof$r -- python $P/tools/one_step.py 20 small_VGG9_cl_128_128 > $P/$O/prof$r.log 2>&1 )
  f=$(find $O/prof$r -name "*kernel_stats.csv" | head -1)
  echo "rows=$r"; grep -E "fc_tail|fc_bwd_combo|gemm_mfma" "$f" | cut -c1-160
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows if not r["Name"].startswith(("__amd","void at::")))
print("kernel time per pass: %.1f us" % (tot/20/1e3))
PY
  rm -rf $O/prof$r
done
SECONDS=0
timeout 1200 python bench.py --sweep-only --sweep-tasks 0 > $O/sweep_only.json 2> $O/sweep_only.err
echo "pair+chain: $SECONDS s rc $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06b2/sweep_only.json").read().strip().splitlines()[-1])
c=d["chain"]; print({k:c[k] for k in ("tasks_compared","max_gap_points","max_omega_sum_rel_gap","cpu_threads_per_leg","cpu_legs_pinned_from_logical_cpu")})
print([(e["task"],e["lambda"],round(e["x"],2),e["gap_points"],round(e["cpu_s"],1)) for e in c["per_task"]])
p=d["pair"]; print("pair gpu %.3f cpu %.1f other %.1f gap %s" % (p["gpu_s"],p["cpu_s"],p["cpu_other_threads"]["seconds"],p["max_accuracy_gap_points"]))
PY
