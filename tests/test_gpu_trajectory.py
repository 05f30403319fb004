"""End-to-end agreement where agreement is possible (north_star: "per-task accuracies, forgetting ... match the reference CPU
path"): the EWC task recipe (main_EWC.py:14-76, train_EWC.py:23-86,164-197) on runners that share start model, importance
weights, fresh head and every batch — the HIP path (default kernels: bf16-split convolutions on the large maps, Winograd F(2x2,3x3)
on the other 3x3 layers; with CLHIP_BS=2 / 0: the bf16-split kernels wherever they can run / nowhere; with CLHIP_WINO=0
CLHIP_BS=0: direct f32 MFMA kernels), the fp32 CPU oracle at two thread counts, the fp64 oracle.

Penalised training amplifies rounding (tests/trajectory.py): measured on this recipe, the fp32 CPU runs themselves end 3e-3
(relative l2) away from the fp64 run after 240 steps having been 1e-7 .. 2e-5 away after 20 — and which of the two it is
depends on the thread count, i.e. on whether one ReLU / arg-max near-tie happened to flip early.  The GPU path is held to
that yardstick:
  * one step of arithmetic from the shared start (no amplification yet) lands within 1e-6 of the fp64 step;
  * once the fp32 CPU runs have left the rounding regime (>= 1e-3: amplified rounding, not single roundings), the GPU's distance
    from the fp64 trajectory is at most 2x the largest CPU distance at every checkpoint; before that it is at most 2x what the
    CPU runs reach one checkpoint (20 steps) later — the GPU may enter the amplification phase earlier (its Winograd
    kernels round 16 products of transformed operands where a direct convolution rounds one fma chain; the CPU runs
    themselves differ by 200x after 20 steps depending on the thread count), it may not end up at another size;
  * accuracies — the new task's, and the old task's under the new trunk, i.e. what forgetting is computed from
    (eval.py:146-191) — lie inside the spread of the CPU runs (fp64, fp32 at two thread counts, fp32 from a start displaced by
    1e-6) +- 2 of the 200 validation samples at every checkpoint."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

LAM, LR = 40.0, 2e-3      # a regime where the fp32 CPU runs agree with each other (lr 1e-2: old-task accuracy of two CPU thread
                          # counts differs by up to 17 points at single checkpoints — nothing could be asked of a third runner)
SEP_FACTOR = 2.0
ONE_STEP = 1e-6
LEFT_ROUNDING = 1e-3      # a separation above this is amplified rounding, below it single roundings / a few flipped near-ties
HEAD_START = 1            # checkpoints (x 20 steps) a runner may be ahead of the CPU runs on the way there
SAMPLES = 2


def test_trajectory_separation_vs_fp64():
    import trajectory as T
    prob = T.make_problem()
    threads = sorted({1, min(8, os.cpu_count() or 1)})
    runs = {"fp64": T.run_oracle(prob, torch.float64, threads[-1], LAM, LR)}
    for t in threads:
        runs["cpu_fp32_t%d" % t] = T.run_oracle(prob, torch.float32, t, LAM, LR)
    runs["cpu_fp32_displaced_1e-6"] = T.run_oracle(prob, torch.float32, threads[-1], LAM, LR, perturb=1e-6)
    runs["gpu"] = T.run_gpu(prob, LAM, LR)
    runs["gpu_direct_kernels"] = T.run_gpu_in_subprocess(prob, runs["fp64"], LAM, LR, {"CLHIP_WINO": "0", "CLHIP_BS": "0"})
    # the bf16-split convolutions (csrc/bsconv.hip) on every layer they can run / on none (default: on the large maps only)
    runs["gpu_bs_everywhere"] = T.run_gpu_in_subprocess(prob, runs["fp64"], LAM, LR, {"CLHIP_BS": "2"})
    runs["gpu_bs_nowhere"] = T.run_gpu_in_subprocess(prob, runs["fp64"], LAM, LR, {"CLHIP_BS": "0"})
    text, seps = T.table(runs)
    print(text)
    cpu = [n for n in runs if n.startswith("cpu_")]
    gpus = [n for n in runs if n.startswith("gpu")]
    n_val = len(prob["t2"]["val"])
    steps = [r["step"] for r in runs["fp64"]]
    out = {"lambda": LAM, "lr": LR, "steps": steps, "separation_from_fp64": seps,
           "accuracy_new_task": {n: [r["acc_new"] for r in runs[n]] for n in runs},
           "accuracy_old_task_new_trunk": {n: [r["acc_old"] for r in runs[n]] for n in runs}}
    if os.path.isdir("gpurun_out"):
        with open(os.path.join("gpurun_out", "trajectory_separation.json"), "w") as f:
            json.dump(out, f, indent=1)
    assert steps[0] == 1
    worst = [max(seps[n][k] for n in cpu) for k in range(len(steps))]
    for g in gpus:
        assert seps[g][0] <= ONE_STEP, (g, "one step", seps[g][0], worst[0])
        for k in range(1, len(steps)):
            if worst[k] >= LEFT_ROUNDING:
                bound = SEP_FACTOR * worst[k]
            else:
                bound = SEP_FACTOR * worst[min(k + HEAD_START, len(steps) - 1)]
            assert seps[g][k] <= bound, (g, steps[k], seps[g][k], worst[k], bound)
        for k in range(len(steps)):
            for key in ("acc_new", "acc_old"):
                vals = [runs[n][k][key] for n in cpu] + [runs["fp64"][k][key]]
                lo, hi = min(vals) - float(SAMPLES) / n_val - 1e-9, max(vals) + float(SAMPLES) / n_val + 1e-9
                assert lo <= runs[g][k][key] <= hi, (g, steps[k], key, runs[g][k][key], vals)
    # the run is long enough to have left the transient: the new task sits at the level the data allows, on every runner
    assert all(abs(runs[n][-1]["acc_new"] - runs["fp64"][-1]["acc_new"]) <= float(SAMPLES) / n_val + 1e-9 for n in runs)
