#!/bin/bash
# round 6, session K: the reduced sweep of tests/test_gpu_sweep_paths.py with its forced-path table
set -u
mkdir -p gpurun_out
python - > gpurun_out/r06_k.txt 2> gpurun_out/r06_k.err <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
for sizes, ep in (((2000, 500, 500), 12), ((4000, 1000, 1000), 20)):
    res = bench.full_sweep(0, 0, tasks=4, sizes=sizes, epochs=ep)
    print(sizes, ep, json.dumps(res["conditioning"]))
    for row in res["gpu_stability"]:
        print("  ", row["task"], row["lr"], row["A_ft"], row["omega_max"], [(a["lambda"], round(a["val_acc"], 3), round(a["two_lambda_omega_lr"], 2)) for a in row["attempts"]])
    fp = res["forced_paths"]
    print("  forced:", json.dumps(fp.get("well_conditioned")), fp.get("near_limit_tasks"), fp.get("max_gap_all_tasks_points"), fp.get("error"))
    for e in fp.get("per_task", []):
        print("    ", e["task"], e["lambda"], e["lr"], round(e["x"], 2), e["test_acc"], e["previous_task_test_acc"], e["val_acc"], "%.1e" % e["omega_sum_rel_spread"], e["diverged"])
PY
tail -3 gpurun_out/r06_k.err; cat gpurun_out/r06_k.txt
