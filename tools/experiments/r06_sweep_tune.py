"""Round 6, GPU: which 'blobs' task sequence lets the EWC sweep carry an accuracy claim?

For every candidate (scale s of all amplitudes, class overlap q, coarse grid g, coarse-noise multiplier m) run a T-task EWC sweep
through the driver with the reference's defaults (bench.full_sweep without the CPU pair) and print, per task: the learning rate
phase 1 picked, A_ft, max(Omega) the training was penalised with, and every phase-2 attempt with its lambda, validation
accuracy, threshold and 2 * lambda * max(Omega) * lr (the step of penalised SGD along its stiffest coordinate: < 2 stable
without momentum).  Wanted: no diverged attempt (val_acc 0.0), every accepted training below 1, and still a task whose first
attempt FAILS the threshold.

  python tools/experiments/r06_sweep_tune.py T  s,q,g,m[,n_train[,seed]]  ...      (CLHIP_BS picks the kernel path)

Per sweep it also prints how the attempts sit against the heavy-ball stability limit of the stiffest coordinate, 2 (1 + 0.9) = 3.8:
`law` = attempts on the side of the limit their outcome says (x > 3.8 <=> rejected) / all attempts, `margin` = the smallest
|ln(x / 3.8)| over the attempts (a decision within ~0.15 of the limit can go either way with rounding).
"""
import math
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    tasks = int(sys.argv[1])
    for spec in sys.argv[2:]:
        f = [float(v) for v in spec.split(",")]
        s, q, g, m = f[:4]
        sizes = (8000, 2000, 1000) if len(f) < 5 or not f[4] else (int(f[4]), int(f[4]) // 4, int(f[4]) // 8)
        bench.SWEEP_DATA = {"kind": "blobs", "noise": 0.5 * s, "blobs": {"g": g, "amp": 4.0 * s, "noise_lr": 1.2 * s * m, "q": q},
                            "seed": int(f[5]) if len(f) > 5 else 7}
        try:
            r = bench.full_sweep(0, 0, tasks=tasks, sizes=sizes, lr_grid=os.environ.get("TUNE_LR_GRID"), forced=os.environ.get("TUNE_FORCED", "0") == "1")
        except BaseException as e:     # noqa: BLE001
            print("### %s: FAILED %s: %s" % (spec, type(e).__name__, str(e)[:200]), flush=True)
            continue
        print("### %s  BS=%s  lr_grid=%s  data %s" % (spec, os.environ.get("CLHIP_BS", "default"), os.environ.get("TUNE_LR_GRID", "default"), json.dumps(bench.SWEEP_DATA)), flush=True)
        if "gpu_error" in r:
            print("   ERROR", r["gpu_error"])
            for ln in r.get("gpu_log_tail", [])[-12:]:
                print("     ", ln)
            continue
        print("   %.1f s; first accuracies %s" % (r["gpu_s"], ["%.1f" % a for a in r["gpu_first_accuracies"]]))
        print("   final accuracies %s; avg %.1f forgetting %.1f" % (["%.1f" % a for a in r["gpu_final_accuracies"]], r["gpu_avg_accuracy"], r["gpu_avg_forgetting"]))
        for row in r["gpu_stability"]:
            print("   task %d: lr %g A_ft %.3f omega max %.3g sum %.4g | %s" % (
                row["task"], row["lr"], row["A_ft"], row["omega_max"], row["omega_sum"],
                "  ".join("lam %g val %.3f (thr %.3f) 2lOl %.2g" % (a["lambda"], a["val_acc"], a["threshold"], a["two_lambda_omega_lr"])
                          for a in row["attempts"])))
        att = [a for row in r["gpu_stability"] for a in row["attempts"]]
        ok = sum(1 for a in att if (a["two_lambda_omega_lr"] > 3.8) == (a["val_acc"] < a["threshold"]))
        print("   law %d/%d  margin %.2f  accepted lambdas %s  damaged-but-accepted %s" % (
            ok, len(att), min(abs(math.log(max(a["two_lambda_omega_lr"], 1e-9) / 3.8)) for a in att), r["gpu_accepted_lambda_per_task"],
            [row["task"] for row in r["gpu_stability"] if row["attempts"][-1]["val_acc"] < row["A_ft"] - 0.01]))
        print("   omega max per tensor (last task): %s" % r["gpu_stability"][-1]["omega_max_per_tensor"])
        fp = r.get("forced_paths")
        if fp:
            if "error" in fp:
                print("   forced paths: ERROR", fp["error"])
            else:
                print("   forced paths (%.1f s): near-limit tasks %s; well-conditioned maxima %s; all-task max gap %.2f" % (
                    fp["seconds"], fp["near_limit_tasks"], json.dumps(fp["well_conditioned"]), fp["max_gap_all_tasks_points"]))
                for e in fp["per_task"]:
                    print("     task %d lam %g lr %g x %.2f: test %s prev-task %s val %s  sumOmega rel spread %.2e%s" % (
                        e["task"], e["lambda"], e["lr"], e["x"], e["test_acc"], e["previous_task_test_acc"], e["val_acc"],
                        e["omega_sum_rel_spread"], "  DIVERGED on %s" % e["diverged"] if e["diverged"] else ""))
        print("   conditioning: %s" % json.dumps(r.get("conditioning")))
        print("   phase-1 grid of the last task (lr, val acc): %s" % r["gpu_last_grid"])
        print("", flush=True)


if __name__ == "__main__":
    main()
