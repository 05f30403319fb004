#!/usr/bin/env python
"""ONE FULL TASK of bench.py's 10-task EWC sweep on the host cores, measured (not priced from rates), next to the same task on the
GPU: task 2 of the sweep's own sequence (bench.SWEEP_DATA: 8000 / 2000 / 1000 'blobs' images of 3x64x64, 20 classes) with the
reference's defaults — 5-value LR grid, 70-epoch cap with the count-based LR drop / early stop, batch 200, Fisher pass,
stability decay from lambda = 400 (framework/main.py:52-67, framework_train.py:76-144).

GPU side: the build's driver (SI first-task model, then EWC up to task 2) on cuda:0.
CPU side: the SAME driver with oracle/sweep_ref.py's torch-CPU EWC as the method, run the way the build runs the framework on
several devices — `driver --shard` under torch.distributed (gloo), RANKS processes of THREADS threads each, every rank pinned
to its own cores: the five grid nodes of phase 1 side by side, the stability-decay attempts speculatively side by side
(clsurvey_amd/framework/shard.py).  Both sides start task 2 from the same first-task model (trained on the GPU, outside both
timed regions) and the same task files.

    python tools/cpu_full_task.py [--ranks 5] [--threads 16] [--out gpurun_out/cpu_full_task.json]
TEST / BENCH INFRASTRUCTURE: imports oracle/ (the CPU leg); nothing in clsurvey_amd/ does.
"""
import argparse
import contextlib
import io
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

MODEL = "small_VGG9_cl_128_128"


def common(root, device):
    import bench
    b = bench.SWEEP_DATA["blobs"]
    sizes = os.environ.get("CLHIP_CPUTASK_SIZES", "8000,2000,1000")          # (smoke tests of this script: smaller tasks)
    spec = "2,20,%s,64,%g,%s,%g,%g,%g,%g" % (sizes, bench.SWEEP_DATA["noise"], bench.SWEEP_DATA["kind"], b["g"], b["amp"], b["noise_lr"], b["q"])
    return [MODEL, "--num_epochs", os.environ.get("CLHIP_CPUTASK_EPOCHS", "70"), "--synthetic", spec, "--device", device, "--results_root", root]


def worker(root, threads):
    """one CPU rank (launched by torch.distributed.run): the sharded driver with the oracle's EWC"""
    from clsurvey_amd.framework import driver, shard
    from oracle import sweep_ref
    rank = int(os.environ["RANK"])
    pinned = None
    if hasattr(os, "sched_setaffinity"):
        allowed = sorted(os.sched_getaffinity(0))           # (a container may own a subset of the host's logical CPUs)
        if len(allowed) >= threads * int(os.environ["WORLD_SIZE"]):
            mine = allowed[rank * threads:(rank + 1) * threads]
            try:
                os.sched_setaffinity(0, mine)
                pinned = [mine[0], mine[-1]]
            except OSError:
                pass
    torch.set_num_threads(threads)
    shard.init_from_env("gloo")
    meth = sweep_ref.OracleEWC("small_VGG9")
    t_start = time.perf_counter()

    def progress():                 # what this rank has done so far, every 15 s: a timed-out run still reports a lower bound
        import threading
        with open(os.path.join(root, "progress_rank%d.json" % rank), "w") as f:
            json.dump({"rank": rank, "elapsed_s": time.perf_counter() - t_start, "image_passes": dict(meth.image_passes),
                       "busy_s": dict(meth.seconds)}, f)
        t = threading.Timer(15.0, progress)
        t.daemon = True
        t.start()
    progress()
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            shard.barrier()
            t0 = time.perf_counter()
            out = driver.main(common(root, "cpu") + ["--shard", "--method_name", "EWC", "--max_task_count", "2"], method=meth)
            shard.barrier()
            dt = time.perf_counter() - t0
    except BaseException:
        import traceback
        with open(os.path.join(root, "error_rank%d.txt" % rank), "w") as f:
            f.write(traceback.format_exc())
        raise
    hf = out["frameworks"][-1]
    res = {"rank": rank, "seconds": dt, "pinned_to_logical_cpus": pinned, "image_passes": dict(meth.image_passes), "busy_s": dict(meth.seconds),
           "phase1": [[float(lr), float(a)] for lr, _, a in out["manager"].grid_trace],
           "phase2": [[float(h["lambda"]), float(a), float(th)] for h, a, th in hf.trace]}
    with open(os.path.join(root, "result_rank%d.json" % rank), "w") as f:
        json.dump(res, f)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=5)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "cpu_full_task.json"))
    ap.add_argument("--worker", default=None)
    ap.add_argument("--cpu-timeout", type=float, default=1500.0, help="seconds the CPU ranks may take before the run reports a lower bound")
    a = ap.parse_args()
    if a.worker:
        return worker(a.worker, a.threads)
    from clsurvey_amd.framework import driver
    from clsurvey_amd.methods import method as M
    root = tempfile.mkdtemp(prefix="clhip_cputask_")
    quiet = io.StringIO()
    try:
        groot = os.path.join(root, "gpu")
        with contextlib.redirect_stdout(quiet):
            driver.main(common(groot, "cuda:0") + ["--method_name", "SI", "--runmode", "first_task_basemodel_dump"], method=M.parse("SI"))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gout = driver.main(common(groot, "cuda:0") + ["--method_name", "EWC", "--max_task_count", "2"], method=M.parse("EWC"))
            torch.cuda.synchronize()
            gpu_s = time.perf_counter() - t0
        ghf = gout["frameworks"][-1]
        res = {"what": "task 2 of bench.py's EWC sweep sequence at %s train/val/test 'blobs' images per task (5-LR grid, 70-epoch cap, batch "
                       "200, Fisher pass, stability decay from lambda 400): build's driver on the GPU vs the same driver with the torch-CPU "
                       "oracle's EWC under `--shard` on %d gloo ranks x %d threads"
                       % (os.environ.get("CLHIP_CPUTASK_SIZES", "8000,2000,1000"), a.ranks, a.threads),
               "sizes": os.environ.get("CLHIP_CPUTASK_SIZES", "8000,2000,1000"), "gpu_task_s": gpu_s,
               "gpu_phase1": [[float(lr), float(acc)] for lr, _, acc in gout["manager"].grid_trace],
               "gpu_phase2": [[float(h["lambda"]), float(acc), float(th)] for h, acc, th in ghf.trace]}
        # every CPU rank works in <croot>/rank<r>: give each the task files and the first-task model
        croot = os.path.join(root, "cpu")
        for r in range(a.ranks):
            for sub in ("data", "models", os.path.join("train", "synthetic_tiny_imagenet", "SI")):
                shutil.copytree(os.path.join(groot, sub), os.path.join(croot, "rank%d" % r, sub))
        # (the CPU ranks must not see the GPU: shard.init_from_env would bind rank r to device r)
        env = dict(os.environ, OMP_NUM_THREADS=str(a.threads), MKL_NUM_THREADS=str(a.threads), PYTHONPATH=ROOT,
                   HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
        t0 = time.perf_counter()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.ranks),
               "--master-addr", "127.0.0.1", "--master-port", "29731", os.path.abspath(__file__),
               "--worker", croot, "--threads", str(a.threads)]
        popen = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            out_s, err_s = popen.communicate(timeout=a.cpu_timeout)
        except subprocess.TimeoutExpired:
            import signal
            os.killpg(popen.pid, signal.SIGKILL)         # the process group this script started (torchrun + its ranks), nothing else
            popen.communicate()
            prog = []
            for r in range(a.ranks):
                f = os.path.join(croot, "progress_rank%d.json" % r)
                if os.path.exists(f):
                    prog.append(json.load(open(f)))
            res.update({"cpu_task_s": None, "cpu_timed_out_after_s": a.cpu_timeout, "cpu_progress_at_timeout": prog,
                        "cpu_ranks": a.ranks, "cpu_threads_per_rank": a.threads,
                        "cpu_over_gpu_at_least": a.cpu_timeout / gpu_s})
            os.makedirs(os.path.dirname(a.out), exist_ok=True)
            with open(a.out, "w") as f:
                json.dump(res, f, indent=1)
            print(json.dumps(res))
            return
        proc = subprocess.CompletedProcess(cmd, popen.returncode, out_s, err_s)
        wall = time.perf_counter() - t0
        if proc.returncode != 0:
            errs = ""
            for r in range(a.ranks):
                f = os.path.join(croot, "error_rank%d.txt" % r)
                if os.path.exists(f):
                    errs += "--- rank %d\n%s\n" % (r, open(f).read()[-1500:])
            raise RuntimeError("CPU ranks failed:\n" + (errs or proc.stderr[-3000:]))
        ranks = [json.load(open(os.path.join(croot, "result_rank%d.json" % r))) for r in range(a.ranks)]
        res.update({"cpu_task_s": max(r["seconds"] for r in ranks), "cpu_wall_with_process_start_s": wall,
                    "cpu_ranks": a.ranks, "cpu_threads_per_rank": a.threads, "host_logical_cores": os.cpu_count(),
                    "cpus_this_container_may_use": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
                    "cpu_pinning": [r.get("pinned_to_logical_cpus") for r in ranks],
                    "cpu_image_passes_all_ranks": {k: sum(r["image_passes"][k] for r in ranks) for k in ("train", "eval")},
                    "cpu_phase1": ranks[0]["phase1"], "cpu_phase2": ranks[0]["phase2"],
                    "cpu_over_gpu": max(r["seconds"] for r in ranks) / gpu_s})
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)
        print(json.dumps(res))
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
