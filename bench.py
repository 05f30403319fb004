#!/usr/bin/env python
"""bench.py — EWC small_VGG9 Tiny-ImageNet hot path on MI355X (BASELINE.json configs[1]).

One "step" = one pass of the hot path over one batch of 200 synthetic 3x64x64 images:
  (a) EWC training batch: forward + CE(mean) + backward + Weight_Regularized_SGD.step
      (EWC/train_EWC.py:164-197), and
  (b) importance batch: forward + nll(sum) + backward + Omega += g^2/len
      (diag_fisher, EWC/main_EWC.py:142-156)
i.e. 400 images through forward+backward per step.  Inputs (the whole 8000-image task) are
resident in HBM before the timed region; batches are gathered on device.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      — dominant kernel (by time) vs the fp32-MFMA peak, timed with HIP events on
                  the launch stream inside this process
  cpu_baseline  — the CPU oracle (torch-CPU restatement of the reference path, kind "port")
                  timed on this box's host cores on a bounded sample (rank 0, N=1 only)

Multi-GPU (`--gpus N` under torch.distributed.run): the Continual Hyperparameter Framework's
grid nodes are independent trainings (framework/lr_grid_train.py:51), so every rank runs its
own replica of the workload; no data-path collective; value = all ranks' images / max time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
SMALL = [64, "M", 64, "M", 64, 64, "M", 128, 128, "M"]


def conv_layers(cfg, hw):
    out, c = [], 3
    for v in cfg:
        if v == "M":
            hw //= 2
        else:
            out.append((c, v, hw))
            c = v
    return out


def algorithmic_flops_per_image(cfg, fc, ncls, hw):
    """2*MAC; backward = bwd-data + bwd-weight, no bwd-data for conv1."""
    fwd = 0
    first = None
    for c, k, s in conv_layers(cfg, hw):
        f = 2 * 9 * c * k * s * s
        fwd += f
        if first is None:
            first = f
    last = [v for v in cfg if v != "M"][-1]
    d = last * (hw // 16) ** 2
    for o in list(fc) + [ncls]:
        fwd += 2 * d * o
        d = o
    return fwd, 3 * fwd - first


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=14)      # ~12 s of host work at ~0.85 s per step
    ap.add_argument("--kernel-iters", type=int, default=20)
    return ap.parse_args()


def conv_instance(C, H, W, N, K, mode, pool):
    """Name of the kernel the library launches for this layer (mirrors launch_conv / clhip_conv3x3_relu_pool_fwd in
    csrc/conv3x3.hip), so that the roofline entry can be matched against the rocprofv3 kernel stats in profiles/."""
    if mode == 0 and pool and C == 3 and W % 32 == 0:
        return "conv3x3_c3_relu_pool_kernel"
    kts = (K + 63) // 64
    big = (N * H * W // 128) * kts >= 300
    if W > 16:
        geo = (32, 4, 1) if big else (32, 2, 1)
    elif W > 8:
        geo = (16, 8, 1) if big else (16, 4, 1)
    else:
        geo = (8, 8, 2) if (big and H > 4) else (8, 8, 1)
    ck, vec = (4, "false") if C <= 4 else (8, "true" if (C % 8 == 0 and W % 4 == 0 and W % geo[0] == 0) else "false")
    return "conv3x3_mfma_kernel<%d, %d, %d, %d, %d, %s>" % (geo[0], geo[1], geo[2], ck, mode, vec)


def time_kernels(eng, x, N, iters):
    """Per-layer HIP-event timing of the conv launches of one pass, as the plan executor issues them (fused
    ReLU+pool forward on pooled layers, first-layer weight gradient straight from the pooled gradient)."""
    from clsurvey_amd import ops
    rows = []
    stream = torch.cuda.current_stream()
    cur = x
    for kind, m, relu, pool in eng.layers:
        if kind != "conv":
            break
        C, K = m.in_channels, m.out_channels
        H, W = cur.shape[2], cur.shape[3]
        y = ops.conv3x3_fwd(cur, m.weight.data, m.bias.data, True)
        dy = torch.randn_like(y)
        fl = 2.0 * 9 * C * K * H * W * N

        def timed(fn):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(iters):
                fn()
            e1.record(stream)
            e1.synchronize()
            return e0.elapsed_time(e1) / iters * 1e-3

        xin = cur
        layer = "%dx%d@%d" % (C, K, H)
        if pool:
            yp, idx = ops.conv3x3_relu_pool_fwd(xin, m.weight.data, m.bias.data)
            t_f = timed(lambda: ops.conv3x3_relu_pool_fwd(xin, m.weight.data, m.bias.data))
            rows.append(dict(kernel="conv3x3_relu_pool_fwd", layer=layer, flops=fl, sec=t_f,
                             instance=conv_instance(C, H, W, N, K, 0, True),
                             alg_bytes=4.0 * N * H * W * (C + K / 4.0) + 1.0 * N * K * H * W / 4))
        else:
            t_f = timed(lambda: ops.conv3x3_fwd(xin, m.weight.data, m.bias.data, True))
            rows.append(dict(kernel="conv3x3_fwd", layer=layer, flops=fl, sec=t_f,
                             instance=conv_instance(C, H, W, N, K, 0, False), alg_bytes=4.0 * N * H * W * (C + K)))
        if C == 3 and pool:
            dyp = torch.randn_like(yp)
            t_w = timed(lambda: ops.conv3x3_bwd_weight_unpool(xin, dyp, idx))
            rows.append(dict(kernel="conv3x3_bwd_weight_unpool", layer=layer, flops=fl, sec=t_w,
                             instance="conv3x3_wgrad_smallc_kernel", alg_bytes=4.0 * N * H * W * (C + K / 4.0)))
        else:
            t_w = timed(lambda: ops.conv3x3_bwd_weight(xin, dy))
            rows.append(dict(kernel="conv3x3_bwd_weight", layer=layer, flops=fl, sec=t_w,
                             instance="conv3x3_wgrad_kernel (+ fixed-order reduce)", alg_bytes=4.0 * N * H * W * (C + K)))
        if C > 3:
            t_d = timed(lambda: ops.conv3x3_bwd_data(dy, m.weight.data, xin))
            rows.append(dict(kernel="conv3x3_bwd_data", layer=layer, flops=fl, sec=t_d,
                             instance=conv_instance(K, H, W, N, C, 1, False), alg_bytes=4.0 * N * H * W * (2 * C + K)))
        cur = ops.maxpool2_fwd(y)[0] if pool else y
    return rows


def measured_traffic(kernel, layer, N):
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (tools/gpu_traffic.sh: FETCH_SIZE
    and WRITE_SIZE in separate passes; FETCH_SIZE doubled for 16-byte-per-lane reads as MI355X_MICROARCH.md
    prescribes for gfx950).  None when no measurement for this kernel/shape is on file."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
    except Exception:
        return None
    e = t.get("%s %s N=%d" % (kernel, layer, N))
    return None if e is None else float(e["hbm_bytes_per_launch"])


def cpu_baseline(batch, steps):
    """CPU oracle on the host cores: same step (EWC train batch + Fisher batch)."""
    from oracle import regularizers_ref as R
    from oracle import vgg_ref
    # torch-CPU conv kernels stop scaling (and regress) past a few dozen threads; use the best of a
    # quick probe so the baseline is the CPU's best configuration, and report the threads used.
    ncpu = os.cpu_count() or 1
    gen = np.random.RandomState(7)
    params = vgg_ref.init_params(SMALL, (128, 128), 20, 64, gen)
    omega = [torch.rand_like(p) * 1e-3 for p in params]
    init = [p.clone() for p in params]
    bufs = [None] * len(params)
    fisher = [torch.zeros_like(p) for p in params]
    x = torch.from_numpy(gen.standard_normal((batch, 3, 64, 64)).astype(np.float32))
    y = torch.from_numpy(gen.randint(0, 20, size=(batch,)).astype(np.int64))

    def step(first):
        nonlocal params, bufs, fisher
        _, _, g, _ = vgg_ref.loss_and_grads(params, SMALL, x, y, "ce_mean")
        new = [R.reg_sgd_step(p, gi, o, iv, b, 400, 1e-3, 0.9, 0.0, first)
               for p, gi, o, iv, b in zip(params, g, omega, init, bufs)]
        params, bufs = [n[0] for n in new], [n[1] for n in new]
        _, _, g, _ = vgg_ref.loss_and_grads(params, SMALL, x, y, "ce_sum")
        fisher = [R.fisher_accum(f, gi, 8000) for f, gi in zip(fisher, g)]

    best = None
    for cand in sorted({min(ncpu, c) for c in (16, 32, 64)}):
        torch.set_num_threads(cand)
        step(best is None)           # warm-up at this thread count (first call creates the buffers)
        t0 = time.perf_counter()
        step(False)
        t = time.perf_counter() - t0
        if best is None or t < best[1]:
            best = (cand, t)
    cores = best[0]
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    for _ in range(steps):
        step(False)
    dt = time.perf_counter() - t0
    return dict(value=2 * batch * steps / dt, unit="images/s", cores=cores, kind="port",
                sample="%d steps (EWC train batch + Fisher batch, N=%d) of the torch-CPU oracle, %d threads, %.1f s"
                       % (steps, batch, cores, dt))


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from clsurvey_amd import models, net, ops
    torch.manual_seed(7 + rank)
    N = args.batch
    model = models.parse_model_name("small_VGG9_cl_128_128", (64, 64), 20)
    eng = net.NetEngine(model, N, (3, 64, 64), dev)
    A = eng.arena
    omega = A.buffer("omega")
    omega.uniform_(0, 1e-3)
    init_val = A.buffer("init_val")
    init_val.copy_(A.theta)
    buf = A.buffer("buf")
    fisher = A.buffer("fisher")
    # the whole synthetic task lives in HBM (8000 x 3 x 64 x 64 fp32 = 393 MB)
    g = torch.Generator(device=dev)
    g.manual_seed(7 + rank)
    data_x = torch.randn((8000, 3, 64, 64), generator=g, device=dev)
    data_y = torch.randint(0, 20, (8000,), generator=g, device=dev)
    stats = torch.zeros(2, dtype=torch.float64, device=dev)
    perm = torch.randperm(8000, device=dev)
    nb = 8000 // N

    def step(i, first=False):
        idx = perm[(i % nb) * N:(i % nb + 1) * N]
        x = data_x.index_select(0, idx)
        y = data_y.index_select(0, idx)
        eng.loss_step(x, y, "ce_mean", True, stats)                       # train_EWC.py:181-187
        ops.reg_sgd_step(A.theta, A.grad, omega, init_val, buf, 400.0, 1e-3, 0.9, 0.0, first)   # :189
        j = (i % nb) * N
        eng.loss_step(data_x[j:j + N], data_y[j:j + N], "ce_sum", True)   # main_EWC.py:147-149
        ops.fisher_accum(fisher, A.grad, 8000.0)                          # :155

    step(0, True)
    for i in range(args.warmup):
        step(i + 1)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if not torch.isfinite(A.theta).all():
        raise SystemExit("non-finite parameters after the timed run")

    fwd_fl, step_fl = algorithmic_flops_per_image(SMALL, (128, 128), 20, 64)
    imgs = 2 * N * args.steps * world
    out = {
        "metric": "images/sec (train+importance pass), EWC small_VGG9 Tiny-ImageNet task batch",
        "value": imgs / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "EWC small_VGG9_cl_128_128, Tiny-ImageNet shapes (3x64x64, 20 classes), "
                               "batch 200: train step + Fisher step (BASELINE configs[1])",
                   "images_per_step": 2 * N, "batch": N,
                   "parallelism": "%d independent grid-node replica(s)" % world,
                   "algorithmic_gflop_per_step": 2 * N * step_fl / 1e9,
                   "step_tflops": 2 * N * step_fl * args.steps / dt / 1e12,
                   "step_frac_of_f32_mfma_peak": 2 * N * step_fl * args.steps / dt / 1e12 / PEAK_F32_MFMA_TFLOPS},
    }
    if rank == 0:
        rows = time_kernels(eng, data_x[:N].contiguous(), N, args.kernel_iters)
        agg = {}
        for r in rows:
            a = agg.setdefault(r["kernel"], dict(flops=0.0, sec=0.0, launches=0))
            a["flops"] += r["flops"]; a["sec"] += r["sec"]; a["launches"] += 1
        dom = max(rows, key=lambda r: r["sec"])          # the single launch that costs most per pass
        ach = dom["flops"] / dom["sec"] / 1e12
        out["roofline"] = {"bound": "mfma", "kernel": "%s [%s, layer %s, N=%d]" % (dom["instance"], dom["kernel"], dom["layer"], N),
                           "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS,
                           "traffic": measured_traffic(dom["kernel"], dom["layer"], N),
                           "algorithmic_gflop_per_launch": dom["flops"] / 1e9,
                           "algorithmic_bytes_per_launch": dom["alg_bytes"],
                           "avg_launch_us": dom["sec"] * 1e6,
                           "per_kernel": {k: {"tflops": v["flops"] / v["sec"] / 1e12, "us_per_step_pass": v["sec"] * 1e6}
                                          for k, v in agg.items()},
                           "per_layer": [{"kernel": r["kernel"], "layer": r["layer"], "instance": r["instance"],
                                          "us": r["sec"] * 1e6, "tflops": r["flops"] / r["sec"] / 1e12} for r in rows]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, args.cpu_steps)
            out["config"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()          # rank 0 was still timing kernels: tear the communicator down together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
