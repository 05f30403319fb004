#!/usr/bin/env python
"""bench.py — EWC small_VGG9 Tiny-ImageNet hot path on MI355X (BASELINE.json configs[1]).

One "step" = one pass of the hot path over one batch of 200 synthetic 3x64x64 images:
  (a) EWC training batch: forward + CE(mean) + backward + Weight_Regularized_SGD.step
      (EWC/train_EWC.py:164-197), and
  (b) importance batch: forward + nll(sum) + backward + Omega += g^2/len
      (diag_fisher, EWC/main_EWC.py:142-156)
i.e. 400 images through forward+backward per step.  Inputs (the whole 8000-image task) are
resident in HBM before the timed region; batches are gathered on device.

Prints ONE JSON line (rank 0) as the LAST line of stdout, at most LINE_LIMIT bytes (compact_line): the contract's keys,
`roofline`, `cpu_baseline` and a few scalars.  The full record (everything below) is written to
gpurun_out/bench_details.json and to stderr as `bench-details: {...}`.  Objects of the full record:
  roofline      — dominant kernel (by time) vs the fp32-MFMA peak, timed with HIP events on
                  the launch stream inside this process
  cpu_baseline  — the CPU oracle (torch-CPU restatement of the reference path, kind "port")
                  timed on this box's host cores on a bounded sample (rank 0, N=1 only)

  configs       — the other BASELINE.json configs, each a short measured loop of ITS hot path on this GPU
                  (MAS importance + SI step on base_VGG9_cl_512_512, PackNet batch + HAT step on wide_VGG9 at 64x64 and
                  224x224, GEM observe on AlexNet 224x224 at 1 / 5 / 9 tasks in memory): ms per step, images/s, TFLOP/s
  sweep         — BASELINE.json's second metric on a BOUNDED sweep, run twice with the same task files, LR grid,
                  epoch cap and decay rule: through the build's driver on the GPU and through the same driver with
                  the CPU oracle's EWC method on the host cores (oracle/sweep_ref.py); wall-clock of both, measured

Multi-GPU (`--gpus N` under torch.distributed.run): the phase-1 LR grid of the Continual Hyperparameter Framework is
N independent trainings from one start model (framework/lr_grid_train.py:51-151).  Rank r runs grid node r (LR =
lr_grid[r % 5], iteration r // 5) for the K timed steps; the timed region also holds what the sharded grid exchanges
over RCCL: the broadcast of the start model's parameter arena before the steps, the all_gather of the nodes' metrics
and the broadcast of the winner's arena after them.  No collective on the per-batch path; value = all ranks' images /
max time; `grid.fill_factor` = busy fraction of the GPUs for the reference's default 5-LR single-iteration grid.
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
WINO_ISSUE = 16.0 / 36.0        # F(2x2, 3x3): 16 multiplies per 2x2 output tile and channel pair where the direct form has 36
PEAK_HBM_TBPS = 8.0            # MI355X_MICROARCH.md: HBM3E
SMALL = [64, "M", 64, "M", 64, 64, "M", 128, 128, "M"]
# what the path computes in: fp32 data end to end; the forward / backward-data convolutions of the >= 32-channel layers multiply
# fp32 operands as three bf16 pieces each on the bf16 matrix cores (six products, fp32 accumulation — error of an fp32 chain,
# tests/test_gpu_bs.py::test_bs_error_is_that_of_an_fp32_chain); every other kernel is f32 MFMA / f32 VALU, reductions in f64
DTYPE = "f32 (bf16x6 split operands on the conv fwd/bwd-data matrix products, fp32 accumulate)"


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


def shard_fill(n_nodes, world):
    from clsurvey_amd.framework.shard import fill_factor
    return fill_factor(n_nodes, world)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=14)      # ~12 s of host work at ~0.85 s per step
    ap.add_argument("--kernel-iters", type=int, default=20)
    ap.add_argument("--config-steps", type=int, default=8, help="timed steps per entry of the `configs` object")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` object (BASELINE configs 3-5)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the full 10-task EWC sweep and its GPU / CPU pair")
    ap.add_argument("--sweep-tasks", type=int, default=10)
    ap.add_argument("--sweep-epochs", type=int, default=70)
    ap.add_argument("--sweep-only", action="store_true", help="print only the `sweep` object (tuning the synthetic sequence)")
    ap.add_argument("--pair-cpu-leg", type=str, default=None, help="internal: run one CPU leg of the sweep pair under this root")
    ap.add_argument("--pair-threads", type=int, default=16)
    ap.add_argument("--pair-pin", type=int, default=-1, help="internal: first logical CPU of this leg's affinity set (-1: not pinned)")
    ap.add_argument("--chain-cpu-leg", type=str, default=None, help="internal: run one task of this chain plan (JSON) on the torch-CPU oracle")
    ap.add_argument("--chain-job", type=int, default=0)
    ap.add_argument("--sweep-blobs", type=str, default=None, help="tuning: g,amp,noise_lr,q of the sweep's synthetic tasks")
    ap.add_argument("--forced-leg", type=str, default=None, help="internal: run the teacher-forced trainings of this plan file (JSON) on the "
                                                                 "kernel path CLHIP_BS selects")
    return ap.parse_args()


def conv_instance(C, H, W, N, K, mode, pool, unpool=False):
    """Name of the kernel the library launches for this layer (mirrors launch_conv / clhip_conv3x3_relu_pool_fwd in
    csrc/conv3x3.hip), so that the roofline entry can be matched against the rocprofv3 kernel stats in profiles/."""
    if mode == 0 and pool and C == 3 and W % 64 == 0 and H % 2 == 0:
        return "conv3x3_c3w64_relu_pool_kernel<%s>" % ("true" if K % 64 == 0 else "false")
    if mode == 0 and pool and C == 3 and W % 32 == 0:
        return "conv3x3_c3_relu_pool_kernel"
    kts = (K + 63) // 64
    big = (N * H * W // 128) * kts >= 300
    ck, vec = (4, False) if C <= 4 else (8, C % 8 == 0 and W % 4 == 0 and W % (32 if W > 16 else (16 if W > 8 else 8)) == 0)
    tail = "%d, %d, %s, %s>" % (ck, mode, "true" if vec else "false", "true" if unpool else "false")
    if H == 13 and W == 13 and not vec and not pool:
        return "conv3x3_mfma_kernel<13, 13, 1, " + tail
    if W > 16 and big and vec:
        total = ((W + 31) // 32) * ((H + 3) // 4) * kts * N
        if total > 256 and 0 < total % 256 <= 192:          # whole rounds of 128-pixel tiles + the odd images as 64-pixel tiles
            return "conv3x3_mfma_mixed_kernel<32, 4, 2, " + tail
    if W <= 16 and big and H > 4 and vec:
        w16 = W > 8
        nba = 1 if w16 else 2
        per_unit = ((W + (15 if w16 else 7)) // (16 if w16 else 8)) * ((H + 7) // 8) * kts
        total = per_unit * ((N + nba - 1) // nba)
        if total >= 512 and 0 < total % 256 <= 192 and N % nba == 0:
            return ("conv3x3_mfma_mixed2_kernel<16, 8, 1, 4, 1, " if w16 else "conv3x3_mfma_mixed2_kernel<8, 8, 2, 8, 1, ") + tail
    if W > 16:
        geo = (32, 4, 1) if big else (32, 2, 1)
    elif W > 8:
        geo = (16, 8, 1) if big else (16, 4, 1)
    else:
        geo = (8, 8, 2) if (big and H > 4) else (8, 8, 1)
    return "conv3x3_mfma_kernel<%d, %d, %d, " % geo + tail


def wino_conv_instance(W, mode, unpool, N=None, kout=None):
    """Instance name of the Winograd forward (mode 0) / backward-data (mode 1) launch for a W-wide even map (csrc/wino.hip, launch_wino)."""
    if W == 8 and N is not None and ((N * 16 + 31) // 32) * ((kout + 31) // 32) < 640:
        return "wino_conv16_kernel<%d, %s, %d> (+ wino_weight_kernel)" % (mode, "true" if unpool else "false", 1 if N * ((kout + 31) // 32) <= 512 else 2)
    return "wino_conv16g_kernel<%s, %d, %s> (+ wino_weight_kernel)" % ("8, 2, 4" if W >= 16 else "4, 4, 4", mode, "true" if unpool else "false")


PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16)
BS_ISSUE = 6.0                     # bf16-split direct convolution: six bf16 MFMA products per fp32 multiply (csrc/bsconv.hip)


def bs_conv_instance(N, H, W, kout, mode, unpool):
    """Instance name of the bf16-split forward (mode 0) / backward-data (mode 1) launch (csrc/bsconv.hip, bs_launch)."""
    geo = "32, 4, 1, 2, 2, 1, 3" if W > 16 else ("16, 8, 1, 2, 2, 1, 3" if W > 8 else "8, 8, 2, 2, 2, 1, 3")
    u = "true" if unpool else "false"
    if W > 16 and not ((H | W) & 1) and os.environ.get("CLHIP_BS_MIXED", "1") != "0":
        # whole rounds of 128-pixel blocks + a remainder of at most half a round: the last images go out as 64-pixel tiles in the same
        # grid (bs_launch_mixed32)
        kts = (kout + 63) // 64
        per_img = ((W + 31) // 32) * ((H + 3) // 4)
        slots = 256 * 3
        total = per_img * N * kts
        rounds, rem = divmod(total, slots)
        na = rounds * slots // (per_img * kts)
        if rounds >= 1 and 0 < rem and 2 * rem <= slots and 1 <= na < N and na % 8 == 0:
            return "bs_conv_mixed_kernel<BsGeo<%s>, BsGeo<32, 2, 1, 2, 1, 1, 3>, %d, %s> (+ bs_weight_multi_kernel)" % (geo, mode, u)
    return "bs_conv_kernel<BsGeo<%s>, %d, %s> (+ bs_weight_multi_kernel)" % (geo, mode, u)


def pipe_seconds(flops, path):
    """Seconds the matrix pipe needs for a launch at its peak: direct f32 MFMA: flops / 157.3 T; Winograd F(2x2,3x3) on f32 MFMA:
    16/36 of the multiplies; bf16-split: 6 bf16 products per multiply on the 2.5 PFLOP/s dense bf16 pipe."""
    if path == "bs":
        return flops * BS_ISSUE / (PEAK_BF16_MFMA_TFLOPS * 1e12)
    if path == "wino":
        return flops * WINO_ISSUE / (PEAK_F32_MFMA_TFLOPS * 1e12)
    return flops / (PEAK_F32_MFMA_TFLOPS * 1e12)


def wino_wgrad_instance(N, C, K, H, W, unpool):
    """Instance name of the Winograd weight-gradient launch (csrc/wino.hip, clhip_internal_wino_wgrad_partial): layers with fewer
    than 16 sixteen-tile stages per 64 x 64-tile block take the 32 x 32-tile pixel-split kernel."""
    wide = W >= 16
    tcs, trs = (8, 2) if wide else (4, 4)
    total = ((W // 2 + tcs - 1) // tcs) * ((H // 2 + trs - 1) // trs) * N
    kc = (K // 64) * (C // 64)
    splits = min(1 if kc >= 256 else 256 // kc, total)
    u = "true" if unpool else "false"
    vec = "true" if W % (2 * tcs) == 0 else "false"       # stages staged in 16-byte pieces: maps of whole tiles in width (16-byte-aligned tensors)
    if total < 16 * splits:
        return "wino_wgrad_ps_kernel<%d, %d, %s, %s> (slabs + reduction)" % (tcs, trs, u, vec)
    return "wino_wgrad_kernel<%d, %d, 1, %s, %s> (slabs + reduction)" % (tcs, trs, u, vec)


def time_kernels(eng, x, N, iters):
    """Per-layer HIP-event timing of the conv launches of one pass, as the plan executor issues them (fused
    ReLU+pool forward on pooled layers, first-layer weight gradient straight from the pooled gradient)."""
    from clsurvey_amd import ops
    rows = []
    stream = torch.cuda.current_stream()
    cur = x
    for li, (kind, m, relu, pool) in enumerate(eng.layers):
        if kind != "conv":
            break
        paths = eng.layer_paths(li)          # the kernels the plan executor launches for this layer (Winograd or direct)
        C, K = m.in_channels, m.out_channels
        H, W = cur.shape[2], cur.shape[3]
        y = ops.conv3x3_fwd(cur, m.weight.data, m.bias.data, True)
        dy = torch.randn_like(y)
        fl = 2.0 * 9 * C * K * H * W * N

        def timed(fn):
            fn()
            torch.cuda.synchronize()
            best = float("inf")         # three batches, the fastest one: a one-off host stall (tens of ms, seen on fresh boxes) stays out
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(iters):
                    fn()
                e1.record(stream)
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1) / iters * 1e-3)
            return best

        xin = cur
        # ReLU mask of backward-data: not read when the input is the pooled output of a fused conv + ReLU + pool launch (its
        # arg-max bytes carry the dead windows, csrc/common.hpp) — as the plan executor calls the kernels
        xmask = None if (li > 0 and eng.layers[li - 1][3]) else xin
        layer = "%dx%d@%d" % (C, K, H)
        wt, bs = m.weight.data, m.bias.data
        pf = "bs" if paths.get("bs_fwd") else ("wino" if paths["fwd"] else "direct")              # path of the forward launch
        pd = "bs" if paths.get("bs_bwd_data") else ("wino" if paths["bwd_data"] else "direct")    # ... of backward-data
        pw = "bs" if paths.get("bs_bwd_weight") else ("wino" if paths["bwd_weight"] else "direct")

        def row(kernel, kind, sec, path, instance, alg_bytes):
            rows.append(dict(kernel=kernel, layer=layer, li=li, kind=kind, flops=fl, sec=sec, path=path, winograd=path == "wino",
                             pipe_sec=pipe_seconds(fl, path), instance=instance, alg_bytes=alg_bytes))
        fwd_fn = {"bs": ops.conv3x3_bs_fwd, "wino": ops.conv3x3_wino_fwd}
        if pool:
            yp, idx = ops.conv3x3_relu_pool_fwd(xin, wt, bs)
            t_f = timed((lambda: fwd_fn[pf](xin, wt, bs, True, pool=True)) if pf != "direct" else (lambda: ops.conv3x3_relu_pool_fwd(xin, wt, bs)))
            row("conv3x3_relu_pool_fwd", "fwd", t_f, pf,
                bs_conv_instance(N, H, W, K, 0, False) if pf == "bs" else wino_conv_instance(W, 0, False, N, K) if pf == "wino"
                else conv_instance(C, H, W, N, K, 0, True), 4.0 * N * H * W * (C + K / 4.0) + 1.0 * N * K * H * W / 4)
        else:
            t_f = timed((lambda: fwd_fn[pf](xin, wt, bs, True)) if pf != "direct" else (lambda: ops.conv3x3_fwd(xin, wt, bs, True)))
            row("conv3x3_fwd", "fwd", t_f, pf,
                bs_conv_instance(N, H, W, K, 0, False) if pf == "bs" else wino_conv_instance(W, 0, False, N, K) if pf == "wino"
                else conv_instance(C, H, W, N, K, 0, False), 4.0 * N * H * W * (C + K))
        if pool:
            dyp = torch.randn_like(yp)
        # the slab kernel alone, as the plan executor launches it (the slabs of all layers are reduced by ONE
        # wgrad_reduce_multi launch at the end of backward: its row is in profiles/*kernel_stats.csv); on pooled layers
        # both backward kernels take the POOLED gradient + arg-max bytes and rebuild the un-pooled tile while staging
        if C == 3 and pool:
            t_w = timed(lambda: ops.conv3x3_bwd_weight_slabs(xin, dyp, idx))
            row("conv3x3_bwd_weight_unpool", "bwd_weight", t_w, "direct",
                "conv3x3_wgrad_c3_unpool_kernel" if W % 32 == 0 else "conv3x3_wgrad_smallc_kernel", 4.0 * N * H * W * (C + K / 4.0))
        elif pool:
            # (the Winograd entry point times slabs + its own reduction launch; inside a pass the reduction is deferred)
            t_w = timed((lambda: ops.conv3x3_bs_bwd_weight(xin, dyp, idx)) if pw == "bs" else
                        (lambda: ops.conv3x3_wino_bwd_weight(xin, dyp, idx)) if pw == "wino" else
                        (lambda: ops.conv3x3_bwd_weight_slabs(xin, dyp, idx)))
            row("conv3x3_bwd_weight_unpool", "bwd_weight", t_w, pw,
                "bs_wgrad_kernel<true> (slabs + reduction)" if pw == "bs" else
                wino_wgrad_instance(N, C, K, H, W, True) if pw == "wino" else "conv3x3_wgrad_kernel<..., UNPOOL=true> (slabs; reduction deferred)",
                4.0 * N * H * W * (C + K / 4.0) + 1.0 * N * K * H * W / 4)
        else:
            t_w = timed((lambda: ops.conv3x3_bs_bwd_weight(xin, dy)) if pw == "bs" else
                        (lambda: ops.conv3x3_wino_bwd_weight(xin, dy)) if pw == "wino" else
                        (lambda: ops.conv3x3_bwd_weight_slabs(xin, dy)))
            row("conv3x3_bwd_weight", "bwd_weight", t_w, pw,
                "bs_wgrad_kernel<false> (slabs + reduction)" if pw == "bs" else
                wino_wgrad_instance(N, C, K, H, W, False) if pw == "wino" else "conv3x3_wgrad_kernel (slabs; reduction deferred)",
                4.0 * N * H * W * (C + K))
        bwd_fn = {"bs": ops.conv3x3_bs_bwd_data, "wino": ops.conv3x3_wino_bwd_data}
        if C > 3 and pool:
            t_d = timed((lambda: bwd_fn[pd](dyp, wt, xmask, idx)) if pd != "direct" else (lambda: ops.conv3x3_bwd_data_unpool(dyp, idx, wt, xmask)))
            row("conv3x3_bwd_data_unpool", "bwd_data", t_d, pd,
                bs_conv_instance(N, H, W, C, 1, True) if pd == "bs" else wino_conv_instance(W, 1, True, N, C) if pd == "wino"
                else conv_instance(K, H, W, N, C, 1, False, True),
                4.0 * N * H * W * ((2 if xmask is not None else 1) * C + K / 4.0) + 1.0 * N * K * H * W / 4)
        elif C > 3:
            t_d = timed((lambda: bwd_fn[pd](dy, wt, xmask)) if pd != "direct" else (lambda: ops.conv3x3_bwd_data(dy, wt, xmask)))
            row("conv3x3_bwd_data", "bwd_data", t_d, pd,
                bs_conv_instance(N, H, W, C, 1, False) if pd == "bs" else wino_conv_instance(W, 1, False, N, C) if pd == "wino"
                else conv_instance(K, H, W, N, C, 1, False), 4.0 * N * H * W * ((2 if xmask is not None else 1) * C + K))
        # layers whose two backward launches the plan executor issues as ONE grid (csrc/wino.hip, wino_pair_kernel): the rows above are
        # the launches of their own; the merged operator (with its own weight transform and slab reduction, which a pass shares
        # between the layers) is timed beside them
        if C > 3 and pd == "wino" and pw == "wino":
            args = (xin, dyp, wt, xmask, idx) if pool else (xin, dy, wt, xmask, None)
            if ops.conv3x3_wino_bwd(*args) is not None:
                t_m = timed(lambda: ops.conv3x3_wino_bwd(*args))
                for r in rows[-2:]:
                    r["one_grid"] = True
                    r["one_grid_us_both_launches"] = t_m * 1e6
        cur = ops.maxpool2_fwd(y)[0] if pool else y
    return rows


def measured_traffic(kernel, layer, N, instance=None):
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
    if e is None or (instance is not None and e.get("instance") != instance):
        return None                 # measured for another kernel instance (tiling changed since): not this launch's traffic
    return float(e["hbm_bytes_per_launch"])


CFGS = {"small_VGG9": SMALL, "base_VGG9": [64, "M", 64, "M", 128, 128, "M", 256, 256, "M"],
        "wide_VGG9": [64, "M", 128, "M", 256, 256, "M", 512, 512, "M"]}


def _timed_loop(fn, steps, warm=2):
    """ms per call: `warm` untimed calls, then `steps` calls between two synchronisations."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def module_flops_per_image(model, hw):
    """2*MAC of every Conv2d / Linear of a features -> classifier net at hw x hw inputs (pool / stride geometry followed)."""
    fwd, first, h = 0.0, None, hw
    for m in model.features.children():
        if isinstance(m, torch.nn.Conv2d):
            h = (h + 2 * m.padding[0] - m.kernel_size[0]) // m.stride[0] + 1
            f = 2.0 * m.kernel_size[0] * m.kernel_size[1] * m.in_channels * m.out_channels * h * h
            fwd += f
            first = f if first is None else first
        elif isinstance(m, torch.nn.MaxPool2d):
            k = m.kernel_size if isinstance(m.kernel_size, int) else m.kernel_size[0]
            st = m.stride if isinstance(m.stride, int) else m.stride[0]
            h = (h - k) // st + 1
    for m in model.classifier.children():
        if isinstance(m, torch.nn.Linear):
            fwd += 2.0 * m.in_features * m.out_features
    return fwd, 3 * fwd - first


def mfma_busy_pmc(width, instance=None):
    """Matrix-pipe busy fraction by PMC (SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES), tools/gpu_mfma_util.sh) from the
    newest profiles/rNN_mfma_util_<width>.csv on file: {"file", "kernels": {instance: busy}} — the entry of `instance` alone
    when given; None when no profile is on file.  Read from the committed profile, not measured in this run."""
    import glob
    key = {"small_VGG9": "small", "base_VGG9": "base", "wide_VGG9": "wide"}.get(width, width)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_mfma_util_%s.csv" % key)))      # (r06 < r06b < r06c: the newest sorts last)
    if not files:
        return None
    rows = {}
    try:
        with open(files[-1]) as f:
            for line in f.read().splitlines()[1:]:           # kernel,dispatches,avg_us,mfma_busy,valu_active (template names hold commas)
                k, _, _, busy, _ = line.rsplit(",", 4)
                rows[k] = float(busy)
    except Exception:
        return None
    name = os.path.basename(files[-1])
    if instance is not None:
        for k, v in rows.items():
            if instance.startswith(k) or k.startswith(instance.split(" (")[0]):
                return {"file": name, "kernel": k, "busy": v}
        return None
    return {"file": name, "kernels": rows}



def _weighted_busy(rows):
    """MFMA-pipe busy (PMC, from the newest profiles/rNN_mfma_util_small.csv) of a group of launches, weighted by their time."""
    num = den = 0.0
    for r in rows:
        b = (mfma_busy_pmc("small_VGG9", r["instance"]) or {}).get("busy")
        if b is not None:
            num += b * r["sec"]
            den += r["sec"]
    return num / den if den > 0 else None


def step_flops_per_image(eng, hw):
    """(algorithmic 2*MAC, matrix-pipe seconds at peak) of one training / importance step per image — forward + backward-data (not on
    the first layer) + weight gradient of every layer of the engine's plan, each launch priced on the path the plan runs it through
    (clhip_net_layer_paths, pipe_seconds).  pipe seconds / measured time is a fraction of the hardware's matrix peak (<= 1 by
    construction); algorithmic / time / f32 peak is the SURVEY 8(d) figure and may exceed 1 on a net whose layers run Winograd or
    on the bf16 pipe."""
    alg = pipe = 0.0
    h = hw
    first = True
    for li, (kind, m, relu, pool) in enumerate(eng.layers):
        if kind == "conv":
            ks, st, pd = m.kernel_size[0], m.stride[0], m.padding[0]
            h = (h + 2 * pd - ks) // st + 1
            f = 2.0 * ks * ks * m.in_channels * m.out_channels * h * h
            paths = eng.layer_paths(li)
            for kind_, bs_key, on in (("fwd", "bs_fwd", True), ("bwd_data", "bs_bwd_data", not first), ("bwd_weight", "bs_bwd_weight", True)):
                if on:
                    alg += f
                    pipe += pipe_seconds(f, "bs" if (bs_key and paths.get(bs_key)) else ("wino" if paths[kind_] else "direct"))
            first = False
            if pool:
                pk, ps = pool if isinstance(pool, tuple) else (2, 2)
                h = (h - pk) // ps + 1
        else:
            f = 2.0 * m.in_features * m.out_features
            alg += 3 * f
            pipe += pipe_seconds(3 * f, "direct")
    return alg, pipe


def hbm_kernels(dev, n=57_823_240, iters=10):
    """The HBM-bound regulariser / optimizer / gradient-memory kernels on an AlexNet-sized parameter arena (57.8 M floats:
    BASELINE configs[3]'s model; on the 0.6-9 M-parameter VGG9s the same launches are latency-bound): achieved TB/s =
    algorithmic bytes per parameter (SURVEY 8d / DESIGN 4) x n / HIP-event time, and its fraction of the 8 TB/s HBM3E peak
    (MI355X_MICROARCH.md)."""
    import ctypes as C
    from clsurvey_amd import _lib, ops
    from clsurvey_amd.methods import packnet as PK
    t = {k: torch.rand(n, device=dev) * 1e-2 for k in ("theta", "grad", "omega", "init", "buf", "w", "out")}
    G = torch.randn((11, n), device=dev)                  # GEM gradient memory: 10 tasks + the current gradient (config 4 at its deepest)
    mask = torch.randint(1, 3, (n,), device=dev, dtype=torch.int64).to(torch.uint8)
    L = _lib.lib()
    ws = torch.zeros(L.clhip_gem_gram_ws(16), dtype=torch.uint8, device=dev)
    gram = torch.zeros(256, dtype=torch.float64, device=dev)
    rows5 = (C.c_int * 5)(0, 1, 2, 3, 4)
    rows11 = (C.c_int * 11)(*range(11))
    v10 = (C.c_float * 10)(*[0.5 + 0.1 * i for i in range(10)])
    v5 = (C.c_float * 5)(0.5, 1.0, -0.5, 0.25, 2.0)
    st = torch.cuda.current_stream().cuda_stream
    cases = [
        ("reg_sgd_step", "EWC / MAS Weight_Regularized_SGD.step (train_EWC.py:23-86)", 28,
         lambda: ops.reg_sgd_step(t["theta"], t["grad"], t["omega"], t["init"], t["buf"], 400.0, 1e-3, 0.9, 0.0, False)),
        ("fisher_accum", "diag_fisher Omega += g^2 / len (main_EWC.py:155)", 12, lambda: ops.fisher_accum(t["omega"], t["grad"], 8000.0)),
        ("mas_accum", "MAS Omega running mean of |g| (train_MAS.py:167-173)", 12, lambda: ops.mas_accum(t["omega"], t["grad"], 3, 200)),
        ("si_step", "Elastic_SGD.step + path integral (train_SI.py:28-126)", 36,
         lambda: ops.si_step(t["theta"], t["grad"], t["omega"], t["init"], t["w"], t["buf"], 400.0, 1e-3, 0.9, 0.0, False)),
        ("si_consolidate", "update_reg_params (train_SI.py:301-364)", 28, lambda: ops.si_consolidate(t["omega"], t["w"], t["theta"], t["init"])),
        ("packnet_sgd_step", "PackNet do_batch tail: foreign grads -> 0, PacknetSGD, pruned -> 0 (packnet/main.py:187-193)", 25,
         lambda: PK.fused_batch_tail(t["theta"], t["grad"], t["buf"], mask, 2, 1e-3, 0.9, 0.0, False)),
        ("gem_store_grad", "GEM store_grad: G[t] = g (gem.py:38-55)", 8,
         lambda: L.clhip_axpy(G[10].data_ptr(), t["grad"].data_ptr(), n, C.c_float(1.0), 1, st)),
        ("gem_gram", "GEM Gram of 5 gradient rows, f64, one pass (gem.py:275-277 + QP inputs)", 20,
         lambda: L.clhip_gem_gram(G.data_ptr(), n, rows5, 5, n, gram.data_ptr(), ws.data_ptr(), ws.numel(), st)),
        ("gem_project", "GEM projection g + sum v_i G_i, 5 rows (gem.py:78-79)", 28,
         lambda: L.clhip_gem_project(G.data_ptr(), n, rows5, v5, 5, t["grad"].data_ptr(), t["out"].data_ptr(), n, st)),
        ("gem_gram_11rows", "GEM Gram of 11 gradient rows (10 tasks in memory + the current gradient: config 4 at its last task), f64, one pass", 44,
         lambda: L.clhip_gem_gram(G.data_ptr(), n, rows11, 11, n, gram.data_ptr(), ws.data_ptr(), ws.numel(), st)),
        ("gem_project_10rows", "GEM projection g + sum v_i G_i, 10 rows (gem.py:78-79)", 48,
         lambda: L.clhip_gem_project(G.data_ptr(), n, rows11, v10, 10, t["grad"].data_ptr(), t["out"].data_ptr(), n, st)),
    ]
    out = []
    for name, what, bpp, fn in cases:
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        out.append({"kernel": name, "what": what, "n_params": n, "algorithmic_bytes_per_param": bpp, "us": us,
                    "achieved_TBps": bpp * n / us / 1e6, "frac_of_hbm_peak": bpp * n / us / 1e6 / PEAK_HBM_TBPS})
    return out


def conv_backward_roofline(dev, N, iters=5):
    """north_star: '>= 40 % MFMA utilisation on the VGG conv backward'.  Per VGG9 width: every backward launch of one pass
    (backward-data and weight-gradient slabs, as the plan executor issues them) timed with HIP events; the dominant one
    and the whole backward as fractions of the fp32-MFMA peak."""
    from clsurvey_amd import models, net
    out = {}
    x = torch.randn((N, 3, 64, 64), device=dev)
    for name, model_name in (("small_VGG9", "small_VGG9_cl_128_128"), ("base_VGG9", "base_VGG9_cl_512_512"),
                             ("wide_VGG9", "wide_VGG9_cl_512_512")):
        eng = net.NetEngine(models.parse_model_name(model_name, (64, 64), 20), N, (3, 64, 64), dev)
        rows = [r for r in time_kernels(eng, x, N, iters) if "bwd" in r["kernel"]]
        dom = max(rows, key=lambda r: r["sec"])

        # `mfma_issued_frac` of a launch = the time its matrix instructions need at the pipe's peak (pipe_seconds: f32 MFMA,
        # Winograd at 16/36, bf16-split at 6 products on the bf16 pipe) / its measured time: a fraction of the hardware peak
        fl, sec = sum(r["flops"] for r in rows), sum(r["sec"] for r in rows)
        out[name] = {"dominant_backward_launch": "%s, layer %s [%s]" % (dom["kernel"], dom["layer"], dom["instance"]),
                     "dominant_path": dom["path"],
                     "dominant_us": dom["sec"] * 1e6, "dominant_algorithmic_tflops": dom["flops"] / dom["sec"] / 1e12,
                     "dominant_mfma_issued_frac": dom["pipe_sec"] / dom["sec"],
                     "all_backward_launches_us": sec * 1e6, "all_backward_algorithmic_tflops": fl / sec / 1e12,
                     "all_backward_algorithmic_over_f32_mfma_peak": fl / sec / 1e12 / PEAK_F32_MFMA_TFLOPS,
                     "all_backward_mfma_issued_frac": sum(r["pipe_sec"] for r in rows) / sec,
                     "worst_backward_launch_mfma_issued_frac": min(r["pipe_sec"] / r["sec"] for r in rows),
                     "paths": {p_: sum(1 for r in rows if r["path"] == p_) for p_ in ("direct", "wino", "bs")},
                     "mfma_busy_pmc": mfma_busy_pmc(name)}
        del eng
    return out


def extra_configs(dev, N, steps):
    """BASELINE.json configs 3-5 on this GPU: a short loop of each config's own hot path (inputs resident in HBM).
    TFLOP/s = algorithmic FLOPs of the step (train / importance pass = 3 x forward - first-layer backward-data) / time."""
    from clsurvey_amd import models, net, ops
    from clsurvey_amd.methods import hat as H
    from clsurvey_amd.methods import packnet as PK
    from clsurvey_amd.methods.gem import GemNet, extend_head
    out = {}
    g = torch.Generator(device=dev)
    g.manual_seed(11)

    def entry(ms, images, flops, what, pipe_s=None):
        """flops: algorithmic; pipe_s: seconds the step's matrix instructions need at the pipes' peaks (step_flops_per_image)."""
        e = {"what": what, "ms_per_step": ms, "images_per_s": images / ms * 1e3, "algorithmic_tflops": flops / ms / 1e9,
             "algorithmic_over_f32_mfma_peak": flops / ms / 1e9 / PEAK_F32_MFMA_TFLOPS}
        if pipe_s is not None:
            e["mfma_issued_frac"] = pipe_s / (ms * 1e-3)
        return e

    # ---- config 3: MAS importance pass + SI step, base_VGG9_cl_512_512, 3x64x64, batch N
    x = torch.randn((N, 3, 64, 64), generator=g, device=dev)
    y = torch.randint(0, 20, (N,), generator=g, device=dev)
    m = models.parse_model_name("base_VGG9_cl_512_512", (64, 64), 20)
    eng = net.NetEngine(m, N, (3, 64, 64), dev)
    A = eng.arena
    step_fl, step_iss = step_flops_per_image(eng, 64)
    omega, init_val, w, buf = A.buffer("omega"), A.buffer("init_val"), A.buffer("w"), A.buffer("buf")
    init_val.copy_(A.theta)
    it = [0]

    def mas():
        eng.loss_step(x, None, "mse_sum_zero", True)                      # train_MAS.py:556-560
        ops.mas_accum(omega, A.grad, it[0], N)                             # :167-173
        it[0] += 1

    def si():
        eng.loss_step(x, y, "ce_mean", True)
        ops.si_step(A.theta, A.grad, omega, init_val, w, buf, 400.0, 1e-3, 0.9, 0.0, it[0] == 0)    # train_SI.py:28-126
        it[0] += 1
    out["mas_importance_base_vgg9"] = entry(_timed_loop(mas, steps), N, N * step_fl,
                                            "MAS compute_importance_l2 batch (fwd + sum(out^2) + bwd + Omega running mean), base_VGG9_cl_512_512 64x64 N=%d" % N,
                                            N * step_iss)
    it[0] = 0
    out["si_step_base_vgg9"] = entry(_timed_loop(si, steps), N, N * step_fl,
                                     "SI training batch (fwd + CE + bwd + Elastic_SGD.step with path integral), base_VGG9_cl_512_512 64x64 N=%d" % N,
                                     N * step_iss)
    del eng, A, omega, init_val, w, buf, m

    # ---- config 5: PackNet batch + HAT step, wide_VGG9_cl_512_512 at 64x64 (N) and 224x224 (iNaturalist geometry, N/4)
    for hw, nb in ((64, N), (224, max(N // 4, 8))):
        x = torch.randn((nb, 3, hw, hw), generator=g, device=dev)
        y = torch.randint(0, 20, (nb,), generator=g, device=dev)
        st = max(2, steps // (4 if hw == 224 else 1))
        m = models.parse_model_name("wide_VGG9_cl_512_512", (hw, hw), 20)
        eng = net.NetEngine(m, nb, (3, hw, hw), dev)
        step_fl, step_iss = step_flops_per_image(eng, hw)
        A = eng.arena
        buf = torch.zeros_like(A.theta)
        mask = torch.randint(1, 3, (A.numel,), generator=g, device=dev, dtype=torch.int64).to(torch.uint8)   # owners 1 / 2
        first = [True]

        def packnet():
            eng.loss_step(x, y, "ce_mean", True)
            PK.fused_batch_tail(A.theta, A.grad, buf, mask, 2, 1e-3, 0.9, 0.0, first[0])      # main.py:187-193
            first[0] = False
        out["packnet_batch_wide_vgg9_%d" % hw] = entry(
            _timed_loop(packnet, st), nb, nb * step_fl,
            "PackNet do_batch (fwd + CE + bwd + foreign-grad zero / PacknetSGD / pruned zero fused), wide_VGG9_cl_512_512 %dx%d N=%d" % (hw, hw, nb),
            nb * step_iss)
        del eng, A, buf, mask
        hn = H.HatNet(m, (3, hw, hw), [(0, 20), (1, 20)]).to(dev)
        hat = H.HatEngine(hn, nb, (3, hw, hw), dev)
        mask_pre, mask_back = H.init_masks(hat, 1, 800.0)
        opt = H.HAT_SGD(hn.parameters(), lr=1e-3, momentum=0.9, weight_decay=0.0)
        count = float(sum(float((1 - mp).sum().item()) for mp in mask_pre))

        def hat_step():
            hat.step(1, x, y, 400.0, mask_pre, 2.5, count, backward=True)          # hat.py:200-249
            opt.step(hn, mask_back, 1, 400.0, 50, 800.0, 10000, thres_emb=6.0)
        out["hat_step_wide_vgg9_%d" % hw] = entry(
            _timed_loop(hat_step, st), nb, nb * step_fl,
            "HAT training batch of task 2 (gates, gated fwd + CE + reg, bwd, HAT_SGD, clamp), wide_VGG9_cl_512_512 %dx%d N=%d" % (hw, hw, nb),
            nb * step_iss)
        del hat, hn, opt, m

    # ---- config 4: GEM observe on AlexNet at 224x224 with 1 / 5 / 9 tasks in memory (mem_per_task 1024, method.py:286)
    nb, n_tasks, nc, mem = N, 10, 20, 1024
    x = torch.randn((nb, 3, 224, 224), generator=g, device=dev)
    m = extend_head(models.AlexNet(num_classes=nc), n_tasks * nc)
    fwd_fl, step_fl = module_flops_per_image(m, 224)
    gem = GemNet(m, n_tasks * nc, n_tasks, [nc] * n_tasks, mem, 1e-3, 0.0, 1.0, batch_size=nb, in_shape=(3, 224, 224), device=dev)
    step_fl, step_iss = step_flops_per_image(gem.engine, 224)
    for past in (1, 5, 9):
        # fill the ring buffers of tasks 0..past-1 with synthetic exemplars, then observe batches of task `past`
        gem.observed_tasks, gem.old_task, gem.mem_cnt = list(range(past)), past - 1, 0
        for t in range(past):
            gem.memory_x[t].normal_(generator=g)
            gem.memory_labels[t].random_(t * nc, (t + 1) * nc, generator=g)
        y = torch.randint(past * nc, (past + 1) * nc, (nb,), generator=g, device=dev)
        ms = _timed_loop(lambda: gem.observe(x, past, y), 2, warm=1)
        imgs = nb + past * mem
        out["gem_observe_alexnet_%dtasks" % past] = entry(
            ms, imgs, imgs * step_fl,
            "GEM observe (%d memory passes of %d exemplars + the batch of %d, Gram, QP, projection, SGD), AlexNet 224x224" % (past, mem, nb),
            imgs * step_iss)
    out["alexnet_mflop_per_image_fwd"] = fwd_fl / 1e6
    del gem, m, x
    torch.cuda.empty_cache()
    out["hbm_kernels"] = hbm_kernels(dev)
    torch.cuda.empty_cache()
    out["conv_backward"] = conv_backward_roofline(dev, N)
    return out


class _PassCounter:
    """Counts the images every DeviceLoader hands out while active, split by loader size: the training split (and the
    Fisher pass over the previous task's training split) is forward + backward, everything smaller is forward only."""

    def __init__(self, n_train):
        from clsurvey_amd import data as D
        self.D, self.n_train, self.counts = D, n_train, {"train": 0, "eval": 0}
        self._orig = D.DeviceLoader.__iter__

    def __enter__(self):
        orig, counts, n_train = self._orig, self.counts, self.n_train

        def counting(loader):
            for x, y in orig(loader):
                counts["train" if loader.n >= n_train else "eval"] += x.shape[0]
                yield x, y
        self.D.DeviceLoader.__iter__ = counting
        return self.counts

    def __exit__(self, *exc):
        self.D.DeviceLoader.__iter__ = self._orig


# ------------------------------------------------------------------------------------------------ sweep + like-for-like pair
# Synthetic task sequence of the sweeps: clsurvey_amd.data.synthetic_task(kind="blobs") — coarse colour patterns with coarse and
# pixel noise and OVERLAPPING classes (q = 0.8: the best possible top-1 accuracy is 81 % whatever the model; measured at
# q = 0.7 the batch-summed Fisher of main_EWC.py:138-157 grows until no lambda of the ten halvings from 400 keeps the
# penalised SGD stable at the grid's learning rates, and the reference's loop ends without a model), so that a trained
# model keeps a non-trivial Fisher diagonal, the stability-decay loop has something to decide, and accuracies saturate at a
# level the data sets.  Models start from torchvision's initialisation (models.py, VGGSlim.py / torchvision VGG: Kaiming
# convolutions, N(0, 0.01) classifier), created by the driver's BaseModel as the reference's models/net.py:158-169 does.
SWEEP_DATA = {"kind": "blobs", "noise": 0.5, "blobs": {"g": 8, "amp": 4.0, "noise_lr": 1.2, "q": 0.8}}
PAIR = {"sizes": (2000, 500, 500), "epochs": 4, "batch": 50, "lr": "1e-2", "lam": 400.0}


def _pair_args(device):
    b = SWEEP_DATA["blobs"]
    spec = "2,20,%d,%d,%d,64,%g,%s,%g,%g,%g,%g" % (tuple(PAIR["sizes"]) + (SWEEP_DATA["noise"], SWEEP_DATA["kind"], b["g"], b["amp"], b["noise_lr"], b["q"]))
    return ["small_VGG9_cl_128_128", "--num_epochs", str(PAIR["epochs"]), "--batch_size", str(PAIR["batch"]), "--saving_freq", "1000",
            "--synthetic", spec, "--device", device, "--lr_grid", PAIR["lr"]]


def _pair_fixed():
    return ["--max_attempts_per_task", "1", "--hyperparams", "%g" % PAIR["lam"], "--method_name", "EWC", "--test"]


def _pair_summary(out, seconds, passes):
    """What both legs report: accuracies of every (task, model) pair, validation accuracies of both phases, omega checksum."""
    last = torch.load(out["model_paths"][-1], map_location="cpu", weights_only=False)
    if hasattr(last, "oracle_omega"):
        osum = float(sum(float(o.double().sum()) for o in last.oracle_omega))
    else:
        osum = float(sum(float(v["omega"].double().sum()) for v in last.reg_params.values() if isinstance(v, dict)))
    return {"seconds": seconds, "image_passes": dict(passes),
            "accuracies": {str(i): [float(a) for a in out["results"][i]["seq_res"][i]] for i in sorted(out["results"])},
            "forgetting": {str(i): [float(a) for a in out["results"][i]["seq_forgetting"][i]] for i in sorted(out["results"])},
            "phase1_val_accuracy": [float(a) for _, _, a in out["manager"].grid_trace],
            "phase2_val_accuracy": [float(a) for _, a, _ in out["frameworks"][-1].trace],
            "trainings_in_phase2": len(out["frameworks"][-1].trace), "omega_sum": osum}


def pair_cpu_leg(root, threads, pin_from=None):
    """One CPU leg of the pair (runs in its own process, `bench.py --pair-cpu-leg ROOT --pair-threads T`): the build's driver
    with oracle/sweep_ref.py's torch-CPU EWC on `threads` host threads, from the first-task model found under ROOT."""
    import contextlib
    import io
    from clsurvey_amd.framework import driver
    from oracle import sweep_ref
    torch.set_num_threads(threads)
    meth = sweep_ref.OracleEWC("small_VGG9")
    pinned = None
    if pin_from is not None and hasattr(os, "sched_setaffinity"):
        try:                                    # its own cores: the other CPU leg runs beside this one
            allowed = sorted(os.sched_getaffinity(0))            # (a container may own a subset of the host's logical CPUs)
            mine = [c for c in allowed if c >= pin_from][:threads]
            if len(mine) == threads:
                os.sched_setaffinity(0, mine)
                pinned = [mine[0], mine[-1]]
        except OSError:
            pass
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        out = driver.main(_pair_args("cpu") + _pair_fixed() + ["--results_root", root], method=meth)
        dt = time.perf_counter() - t0
    res = _pair_summary(out, dt, meth.image_passes)
    res["threads"] = torch.get_num_threads()
    res["pinned_to_logical_cpus"] = pinned
    res["rates_images_per_s"] = {"forward_backward_update": meth.image_passes["train"] / max(meth.seconds["train"], 1e-9),
                                 "forward_only": meth.image_passes["eval"] / max(meth.seconds["eval"], 1e-9)}
    return res


def sweep_stability(out):
    """Per task of a finished EWC sweep: the learning rate phase 1 picked, A_ft, and the step size of penalised SGD along its
    stiffest coordinate, 2 * lambda * max(Omega) * lr, for every phase-2 attempt (train_EWC.py:60-75: the penalty's gradient is
    2 * lambda * Omega * (theta - theta*); plain SGD on that coordinate alone is stable below 2, with momentum 0.9 below 3.8).
    Omega = the importance the task's training was penalised with (sum over the earlier tasks, main_EWC.py:205-232), read from the
    task's own model file."""
    rows = []
    for t, (hf, path) in enumerate(zip(out["frameworks"], out["model_paths"])):
        if hf is None or not hf.trace:
            continue
        m = torch.load(path, map_location="cpu", weights_only=False)
        om = [v["omega"] for v in getattr(m, "reg_params", {}).values() if isinstance(v, dict) and "omega" in v]
        omax = float(max(float(o.max()) for o in om)) if om else 0.0
        lr, a_ft = getattr(hf, "phase1", (None, None))
        rows.append({"task": t + 1, "lr": lr, "A_ft": float(a_ft) if a_ft is not None else None,
                     "omega_max": omax, "omega_max_per_tensor": [float("%.3g" % float(o.max())) for o in om], "omega_sum": float(sum(float(o.double().sum()) for o in om)) if om else 0.0,
                     "attempts": [{"lambda": float(h["lambda"]), "val_acc": float(a), "threshold": float(th),
                                   "two_lambda_omega_lr": 2.0 * float(h["lambda"]) * omax * float(lr) if lr is not None else None}
                                  for h, a, th in hf.trace]})
    return rows


STABILITY_LIMIT = 2.0 * (1.0 + 0.9)      # heavy-ball SGD (momentum 0.9) on a quadratic of curvature h is stable iff lr * h < 2 (1 + 0.9)
NEAR_LIMIT = 3.0                        # a training above this is within 25 % of the limit: not a well-conditioned comparison leg


def sweep_conditioning(rows):
    """How a finished sweep's stability-decay decisions sit against the heavy-ball limit of the stiffest penalised coordinate
    (x = 2 lambda max(Omega) lr; STABILITY_LIMIT = 3.8).  Measured on 40 sweeps of round 6 (profiles/r06_sweep_stability.md):
    every attempt with x < 3.8 met the threshold and every rejected attempt had x > 3.8 (`rejected_all_above_limit`); above the
    limit a training diverges when the stiff coordinate is excited, which depends on rounding — `min_margin` = the smallest
    |ln(x / 3.8)| over the attempts says how close the nearest decision came."""
    import math
    att = [a for row in rows for a in row["attempts"] if a["two_lambda_omega_lr"] is not None]
    if not att:
        return None
    rej = [a for a in att if a["val_acc"] < a["threshold"]]
    return {"attempts": len(att), "rejected": len(rej), "rejected_diverged": sum(1 for a in rej if a["val_acc"] < 0.1),
            "rejected_all_above_limit": all(a["two_lambda_omega_lr"] > STABILITY_LIMIT for a in rej),
            "accepted_above_limit": sum(1 for a in att if a["val_acc"] >= a["threshold"] and a["two_lambda_omega_lr"] > STABILITY_LIMIT),
            "below_limit_rejected": sum(1 for a in rej if a["two_lambda_omega_lr"] <= STABILITY_LIMIT),
            "max_x_accepted": max(a["two_lambda_omega_lr"] for a in att if a["val_acc"] >= a["threshold"]) if len(rej) < len(att) else None,
            "min_margin": min(abs(math.log(max(a["two_lambda_omega_lr"], 1e-12) / STABILITY_LIMIT)) for a in att),
            "limit": STABILITY_LIMIT}


def forced_leg(plan_path, device):
    """One kernel path's leg of the teacher-forced sweep comparison (its own process: CLHIP_BS is read once per process).  For every
    task of the plan: the EWC training of that task exactly as the free-running sweep's accepted attempt ran it — SAME start model
    (the free run's model of the previous task), learning rate and lambda, Fisher pass over the previous task included — from a
    fixed seed; then the new task's test accuracy, the previous task's test accuracy under the new trunk (one-step forgetting),
    and the sum / max of the importance weights the training was penalised with (the previous model's Omega + this path's Fisher)."""
    import contextlib
    import io
    import shutil
    import tempfile
    from clsurvey_amd.framework import driver, inference
    from clsurvey_amd.methods import ewc, method as M
    from clsurvey_amd.methods import train_common as tc
    with open(plan_path) as f:
        plan = json.load(f)
    meth = M.parse("EWC")
    scratch = tempfile.mkdtemp(prefix="clhip_forced_")
    legs = []
    quiet = io.StringIO()
    try:
        for job in plan["jobs"]:
            exp_dir = os.path.join(scratch, "task_%d" % job["task"])
            driver.set_random(7)
            with contextlib.redirect_stdout(quiet):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _, val = ewc.fine_tune_EWC_acuumelation(dataset_path=job["dataset"], previous_task_model_path=job["previous_model"],
                                                        exp_dir=exp_dir, data_dir=None, reg_sets=[job["previous_dataset"]],
                                                        reg_lambda=job["lambda"], num_epochs=plan["epochs"], lr=job["lr"],
                                                        batch_size=plan["batch"], device=device)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = os.path.join(exp_dir, "best_model.pth.tar")
                row = {"task": job["task"], "seconds": dt, "val_acc": float(val), "lr": job["lr"], "lambda": job["lambda"]}
                if os.path.exists(best):
                    model = tc.load_model(best)
                    last = str(len(model.classifier._modules) - 1)
                    om = [v["omega"] for v in model.reg_params.values() if isinstance(v, dict) and "omega" in v]
                    row["omega_sum"] = float(sum(float(o.double().sum()) for o in om))
                    row["omega_max"] = float(max(float(o.max()) for o in om))
                    # (the parameters the NEXT training is penalised on: everything but the head; reg_params also keeps the entries
                    # of the heads of earlier tasks, main_EWC.py:160-232)
                    trunk = list(model.parameters())[:-2]
                    row["omega_sum_trunk"] = float(sum(float(model.reg_params[p]["omega"].double().sum()) for p in trunk if p in model.reg_params))
                    row["test_acc"] = inference.test_model(meth, model, job["dataset"], 0, inference.get_prev_heads(best, last, device),
                                                           batch_size=plan["batch"], device=device)
                    row["previous_task_test_acc"] = inference.test_model(
                        meth, tc.load_model(best), job["previous_dataset"], 0, inference.get_prev_heads(job["previous_model"], last, device),
                        batch_size=plan["batch"], device=device)
                else:
                    row["diverged"] = True                    # train_EWC.py:204-205: the training stopped on a NaN / 1e4 loss
            legs.append(row)
            shutil.rmtree(exp_dir, ignore_errors=True)
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    return {"clhip_bs": os.environ.get("CLHIP_BS", "1"), "tasks": legs}


def forced_paths(out, stability, groot, epochs, batch=200, paths=("0", "1", "2"), condition=True):
    """The three fp32-grade kernel paths (CLHIP_BS = 0: Winograd f32 on every 3x3 layer; 1: the default; 2: bf16-split wherever it
    runs) compared TASK BY TASK on the sweep that just ran: each path repeats every task's accepted training from the free run's
    own previous model, learning rate and lambda (`forced_leg`), in its own process, the three side by side on this GPU.  A free-
    running sweep cannot be compared this way: which stability-decay attempts diverge is decided by rounding (sweep_conditioning),
    on any fp32 implementation.  `condition`: a task whose accepted training sits within 25 % of the stability limit (x > NEAR_LIMIT)
    is repeated at the accepted lambda halved until x <= NEAR_LIMIT (as the chain does), so that every task is a well-conditioned leg.  Reports per task the largest accuracy difference between the paths (new task, previous task under
    the new trunk), the validation accuracies, and the relative spread of Sum(Omega); legs whose training sits within 25 % of the
    stability limit (x > NEAR_LIMIT) are listed and left out of the `well_conditioned` maxima."""
    import subprocess
    jobs = []
    for row in stability:
        t = row["task"]
        if t < 2 or row["lr"] is None:
            continue
        acc = row["attempts"][-1]
        # lambda of the comparison = the accepted one, halved further along the reference's schedule (framework_train.py:168-216) until
        # x = 2 lambda max(Omega) lr <= NEAR_LIMIT: two kernel paths are compared, not two draws of a training at its stability limit
        lam, x = float(acc["lambda"]), acc["two_lambda_omega_lr"]
        halvings = 0
        while condition and x is not None and x > NEAR_LIMIT and halvings < 12:
            lam, x, halvings = 0.5 * lam, 0.5 * x, halvings + 1
        jobs.append({"task": t, "dataset": out["ds_paths"][t - 1], "previous_dataset": out["ds_paths"][t - 2],
                     "previous_model": out["model_paths"][t - 2], "lr": float(row["lr"]), "lambda": lam, "x": x,
                     "lambda_accepted_by_free_run": float(acc["lambda"]), "x_of_accepted": acc["two_lambda_omega_lr"],
                     "halvings_below_accepted": halvings})
    plan_path = os.path.join(groot, "forced_plan.json")
    with open(plan_path, "w") as f:
        json.dump({"jobs": jobs, "epochs": epochs, "batch": batch}, f)
    t0 = time.perf_counter()
    procs = []
    for p in paths:
        env = dict(os.environ, CLHIP_BS=p)
        errf = open(os.path.join(groot, "forced_%s.stderr" % p), "w+")
        procs.append((p, subprocess.Popen([sys.executable, os.path.abspath(__file__), "--forced-leg", plan_path], stdout=subprocess.PIPE,
                                          stderr=errf, env=env, text=True), errf))
    legs = {}
    for p, proc, errf in procs:
        so, _ = proc.communicate(timeout=1200)
        errf.seek(0)
        se = errf.read()
        errf.close()
        line = [ln for ln in so.splitlines() if ln.startswith("{")]
        if proc.returncode != 0 or not line:
            raise RuntimeError("forced leg CLHIP_BS=%s failed:\n%s" % (p, se[-1500:]))
        legs[p] = json.loads(line[-1])["tasks"]
    res = {"what": "every task's accepted EWC training repeated on CLHIP_BS=%s from the free run's previous model, lr and lambda "
                   "(teacher-forced; fixed seed), three processes side by side" % "/".join(paths),
           "seconds": time.perf_counter() - t0, "per_task": [], "near_limit_tasks": []}
    worst = {"test_acc": 0.0, "previous_task_test_acc": 0.0, "val_acc_points": 0.0, "omega_sum_rel": 0.0}
    for i, job in enumerate(jobs):
        rows = [legs[p][i] for p in paths]
        entry = {"task": job["task"], "lr": job["lr"], "lambda": job["lambda"], "x": job["x"],
                 "lambda_accepted_by_free_run": job["lambda_accepted_by_free_run"], "x_of_accepted": job["x_of_accepted"],
                 "diverged": [p for p, r in zip(paths, rows) if r.get("diverged")]}
        gaps = {}
        for key, scale in (("test_acc", 1.0), ("previous_task_test_acc", 1.0), ("val_acc", 100.0)):
            vals = [r[key] * scale for r in rows if key in r]
            entry[key] = [round(v, 3) for v in vals]
            gaps[key] = (max(vals) - min(vals)) if len(vals) == len(rows) else float("inf")
        oms = [r["omega_sum"] for r in rows if "omega_sum" in r]
        entry["omega_sum"] = oms
        gaps["omega_sum_rel"] = (max(oms) - min(oms)) / max(max(oms), 1e-30) if len(oms) == len(rows) else float("inf")
        entry["gap_points"] = {"test_acc": gaps["test_acc"], "previous_task_test_acc": gaps["previous_task_test_acc"], "val_acc": gaps["val_acc"]}
        entry["omega_sum_rel_spread"] = gaps["omega_sum_rel"]
        res["per_task"].append(entry)
        if job["x"] is not None and job["x"] > NEAR_LIMIT:
            res["near_limit_tasks"].append(job["task"])
            continue
        worst["test_acc"] = max(worst["test_acc"], gaps["test_acc"])
        worst["previous_task_test_acc"] = max(worst["previous_task_test_acc"], gaps["previous_task_test_acc"])
        worst["val_acc_points"] = max(worst["val_acc_points"], gaps["val_acc"])
        worst["omega_sum_rel"] = max(worst["omega_sum_rel"], gaps["omega_sum_rel"])
    res["max_gap_all_tasks_points"] = max([max(e["gap_points"]["test_acc"], e["gap_points"]["previous_task_test_acc"]) for e in res["per_task"]] or [0.0])
    res["well_conditioned"] = {"tasks": [j["task"] for j in jobs if not (j["x"] is not None and j["x"] > NEAR_LIMIT)],
                               "max_gap_points_new_task": worst["test_acc"], "max_gap_points_previous_task": worst["previous_task_test_acc"],
                               "max_gap_points_validation": worst["val_acc_points"], "max_omega_sum_rel_spread": worst["omega_sum_rel"]}
    return res


# ------------------------------------------------------------------------------------------------ CPU path vs HIP path, task by task
# A bounded task sequence (PAIR's sizes and schedule, CHAIN["tasks"] tasks) run freely through the driver on the GPU — the reference's
# loop with its LR grid, Fisher pass and stability decay from lambda 400 — and then EVERY task's EWC training repeated on both sides
# from the free run's own model of the task before (teacher-forced, as forced_paths does between kernel paths): the HIP path in this
# process, the torch-CPU oracle (oracle/sweep_ref.py) in one host process per task, all of them side by side (the GPU's 10-task sweep
# has run before them, on an idle host).  lambda of a task's comparison job = the lambda the free run accepted, halved further (the reference's own decay
# schedule, framework_train.py:168-216) until x = 2 lambda max(Omega) lr <= NEAR_LIMIT: above it penalised SGD sits at its stability
# limit and two fp32 implementations may differ by whether the stiff coordinate gets excited (sweep_conditioning).
CHAIN = {"tasks": 4}


def _chain_args(device, root):
    b = SWEEP_DATA["blobs"]
    spec = "%d,20,%d,%d,%d,64,%g,%s,%g,%g,%g,%g" % ((CHAIN["tasks"],) + tuple(PAIR["sizes"]) + (SWEEP_DATA["noise"], SWEEP_DATA["kind"], b["g"], b["amp"], b["noise_lr"], b["q"]))
    return ["small_VGG9_cl_128_128", "--num_epochs", str(PAIR["epochs"]), "--batch_size", str(PAIR["batch"]), "--saving_freq", "1000",
            "--synthetic", spec, "--device", device, "--lr_grid", PAIR["lr"], "--results_root", root]


def chain_cpu_leg(plan_path, index, threads, pin_from=None):
    """One task of the chain on the torch-CPU oracle (its own process: `bench.py --chain-cpu-leg PLAN --chain-job I`): Fisher pass over
    the previous task, omega accumulated onto the previous model's (read from the HIP run's model file: the artefact is the
    reference's own, a pickled nn.Module with reg_params), penalised SGD with the count-based LR drop, best-validation model kept —
    then the new task's test accuracy and the previous task's under the new trunk, as forced_leg reports them for the HIP path."""
    import shutil
    import tempfile
    import types
    from clsurvey_amd.framework import driver
    from oracle import sweep_ref
    torch.set_num_threads(threads)
    if pin_from is not None and hasattr(os, "sched_setaffinity"):
        try:
            mine = [c for c in sorted(os.sched_getaffinity(0)) if c >= pin_from][:threads]
            if len(mine) == threads:
                os.sched_setaffinity(0, mine)
        except OSError:
            pass
    with open(plan_path) as f:
        plan = json.load(f)
    job = plan["jobs"][index]
    meth = sweep_ref.OracleEWC("small_VGG9")
    scratch = tempfile.mkdtemp(prefix="clhip_chain_cpu_")
    try:
        args = types.SimpleNamespace(num_epochs=plan["epochs"], batch_size=plan["batch"], lr=job["lr"], weight_decay=0.0)
        manager = types.SimpleNamespace(current_task_dataset_path=job["dataset"], reg_sets=[job["previous_dataset"]],
                                        previous_task_model_path=job["previous_model"], heuristic_exp_dir=scratch)
        driver.set_random(7)
        t0 = time.perf_counter()
        _, val = meth.train(args, manager, {"lambda": job["lambda"]})
        dt = time.perf_counter() - t0
        row = {"task": job["task"], "seconds": dt, "val_acc": float(val), "lr": job["lr"], "lambda": job["lambda"], "threads": torch.get_num_threads()}
        best = os.path.join(scratch, "best_model.pth.tar")
        if os.path.exists(best):
            model = torch.load(best, map_location="cpu", weights_only=False)
            row["omega_sum_trunk"] = float(sum(float(o.double().sum()) for o in model.oracle_omega[:-2]))
            row["omega_max"] = float(max(float(o.max()) for o in model.oracle_omega[:-2]))

            def acc(dset, head):
                return meth.inference_eval(types.SimpleNamespace(dset_path=dset, test_set="test", eval_model_path=best, head_paths=head,
                                                                 batch_size=plan["batch"]), None)
            row["test_acc"] = acc(job["dataset"], best)
            row["previous_task_test_acc"] = acc(job["previous_dataset"], job["previous_model"])
        else:
            row["diverged"] = True
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    return row


def chain_start(root, dev, cpu_threads, pin_from=None, pin_room=0):
    """GPU: free run of the chain's sequence, then the jobs; starts one CPU process per job and returns the state chain_collect reads.
    The CPU legs share the logical CPUs [pin_from, pin_from + pin_room) in equal parts when given (full_sweep: the free cores of the
    socket the pair's second leg runs on, so that the pair's timed first leg keeps its socket to itself), else run unpinned."""
    import contextlib
    import io
    import subprocess
    from clsurvey_amd.framework import driver
    from clsurvey_amd.methods import method as M
    croot = os.path.join(root, "chain_gpu")
    quiet = io.StringIO()
    with contextlib.redirect_stdout(quiet):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        driver.main(_chain_args(dev, croot) + ["--method_name", "SI", "--runmode", "first_task_basemodel_dump"], method=M.parse("SI"))
        out = driver.main(_chain_args(dev, croot) + ["--method_name", "EWC", "--test"], method=M.parse("EWC"))
        torch.cuda.synchronize()
        free_s = time.perf_counter() - t0
    stab = sweep_stability(out)
    jobs = []
    for row in stab:
        t = row["task"]
        if t < 2 or row["lr"] is None:
            continue
        lam_acc = float(row["attempts"][-1]["lambda"])
        per_lambda = 2.0 * row["omega_max"] * float(row["lr"])
        lam, halvings = lam_acc, 0
        while lam * per_lambda > NEAR_LIMIT and halvings < 12:
            lam *= 0.5
            halvings += 1
        jobs.append({"task": t, "dataset": out["ds_paths"][t - 1], "previous_dataset": out["ds_paths"][t - 2],
                     "previous_model": out["model_paths"][t - 2], "lr": float(row["lr"]), "lambda": lam,
                     "lambda_accepted_by_free_run": lam_acc, "halvings_below_accepted": halvings,
                     "x": lam * per_lambda, "x_of_accepted": lam_acc * per_lambda,
                     "free_run_attempts": [[a["lambda"], a["val_acc"]] for a in row["attempts"]]})
    plan_path = os.path.join(root, "chain_plan.json")
    with open(plan_path, "w") as f:
        json.dump({"jobs": jobs, "epochs": PAIR["epochs"], "batch": PAIR["batch"]}, f)
    procs = []
    share = pin_room // max(len(jobs), 1) if pin_from is not None else 0
    if share >= 4:
        cpu_threads = min(cpu_threads, share)
    for j, job in enumerate(jobs):
        pin = pin_from + j * share if share >= 4 else -1
        env = dict(os.environ, OMP_NUM_THREADS=str(cpu_threads), MKL_NUM_THREADS=str(cpu_threads),
                   HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
        errf = open(os.path.join(root, "chain_cpu_%d.stderr" % j), "w+")
        procs.append((subprocess.Popen([sys.executable, os.path.abspath(__file__), "--chain-cpu-leg", plan_path, "--chain-job", str(j),
                                        "--pair-threads", str(cpu_threads), "--pair-pin", str(pin)],
                                       stdout=subprocess.PIPE, stderr=errf, env=env, text=True), errf))
    r = out["results"]
    free = {"seconds": free_s, "accepted_lambda_per_task": [float(row["attempts"][-1]["lambda"]) for row in stab if row["attempts"]],
            "final_accuracies": [r[i]["seq_res"][i][-1] for i in sorted(r)], "first_accuracies": [r[i]["seq_res"][i][0] for i in sorted(r)],
            "conditioning": sweep_conditioning(stab)}
    # the HIP path's own legs of the same jobs (this process, default kernel path)
    t0 = time.perf_counter()
    gpu_rows = forced_leg(plan_path, dev)["tasks"]
    return {"jobs": jobs, "procs": procs, "gpu_rows": gpu_rows, "gpu_forced_s": time.perf_counter() - t0, "free": free,
            "cpu_threads_per_leg": cpu_threads, "cpu_pinned_from": pin_from if share >= 4 else None}


def chain_collect(state):
    import subprocess
    cpu_rows = []
    for proc, errf in state["procs"]:
        try:
            so, _ = proc.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            proc.kill()
            so, _ = proc.communicate()
        errf.seek(0)
        se = errf.read()
        errf.close()
        line = [ln for ln in so.splitlines() if ln.startswith("{")]
        if proc.returncode != 0 or not line:
            raise RuntimeError("chain CPU leg failed:\n%s" % se[-2000:])
        cpu_rows.append(json.loads(line[-1]))
    state["procs"] = []
    res = {"what": "%d-task sequence (%d/%d/%d images per task, batch %d, %d-epoch cap, LR grid {%s}) run freely on the GPU through the driver "
                   "(stability decay from lambda 400); then every task's EWC training — Fisher pass, omega accumulation, penalised SGD — "
                   "repeated from the free run's model of the task before on the HIP path and on the torch-CPU oracle (one host process "
                   "per task), same task files, start model, seed, batches, head initialisation; lambda = the accepted one, halved "
                   "until x = 2 lambda max(Omega) lr <= %g" % ((CHAIN["tasks"],) + tuple(PAIR["sizes"]) + (PAIR["batch"], PAIR["epochs"], PAIR["lr"], NEAR_LIMIT)),
           "free_run": state["free"], "gpu_forced_seconds": state["gpu_forced_s"], "cpu_threads_per_leg": state["cpu_threads_per_leg"],
           "cpu_legs_pinned_from_logical_cpu": state["cpu_pinned_from"], "per_task": []}
    worst = {"test_acc": 0.0, "previous_task_test_acc": 0.0, "val_acc": 0.0, "omega": 0.0}
    for job, g, c in zip(state["jobs"], state["gpu_rows"], cpu_rows):
        e = {k: job[k] for k in ("task", "lr", "lambda", "lambda_accepted_by_free_run", "halvings_below_accepted", "x", "x_of_accepted")}
        e["diverged"] = [n for n, r in (("gpu", g), ("cpu", c)) if r.get("diverged")]
        e["gpu_s"], e["cpu_s"] = g["seconds"], c["seconds"]
        gaps = {}
        for key, scale in (("test_acc", 1.0), ("previous_task_test_acc", 1.0), ("val_acc", 100.0)):
            if key in g and key in c:
                e[key] = {"gpu": round(g[key] * scale, 3), "cpu": round(c[key] * scale, 3)}
                gaps[key] = abs(g[key] - c[key]) * scale
            else:
                gaps[key] = float("inf")
        e["gap_points"] = gaps
        if "omega_sum_trunk" in g and "omega_sum_trunk" in c:
            e["omega_sum_trunk"] = {"gpu": g["omega_sum_trunk"], "cpu": c["omega_sum_trunk"]}
            e["omega_sum_trunk_rel_gap"] = abs(g["omega_sum_trunk"] - c["omega_sum_trunk"]) / max(abs(c["omega_sum_trunk"]), 1e-30)
        else:
            e["omega_sum_trunk_rel_gap"] = float("inf")
        for k in ("test_acc", "previous_task_test_acc", "val_acc"):
            worst[k] = max(worst[k], gaps[k])
        worst["omega"] = max(worst["omega"], e["omega_sum_trunk_rel_gap"])
        res["per_task"].append(e)
    res["tasks_compared"] = len(res["per_task"])
    res["max_gap_points"] = {"new_task": worst["test_acc"], "previous_task": worst["previous_task_test_acc"], "validation": worst["val_acc"]}
    res["max_omega_sum_rel_gap"] = worst["omega"]
    res["cpu_over_gpu_seconds"] = (sum(e["cpu_s"] for e in res["per_task"]) / max(sum(e["gpu_s"] for e in res["per_task"]), 1e-9)) if res["per_task"] else None
    return res


def full_sweep(dev_index, cpu_threads, tasks=10, sizes=(8000, 2000, 1000), epochs=70, cpu_rates=None, lr_grid=None, forced=True, chain=True):
    """BASELINE.json's second metric ('full-sweep wall-clock'), two measurements.

    `gpu_s`: what framework/main.py runs for `small_VGG9_cl_128_128 --method_name EWC --test` with the reference's defaults
    (5-value LR grid, 70-epoch cap with the count-based LR drop / early stop, batch 200, lambda = 400, drop margin 0.2, decay
    0.5, up to 10 attempts per task) on a `tasks`-task sequence of Tiny-ImageNet's shape (20 classes, 8000 / 2000 / 1000
    images of 3x64x64 per task; SWEEP_DATA above, there is no dataset on the box), through the build's driver on this GPU: SI
    first-task model (main.py:226-241), then per task phase-1 LR grid, Fisher pass, stability decay, and at the end every
    model evaluated on every task (eval.py:146-247).  Measured end to end, task files already written.

    `pair`: ONE bounded task of such a sweep on both sides — task 2 of a PAIR["sizes"] sequence from the first-task model the
    GPU trained (outside every timed region), one learning rate, a PAIR["epochs"]-epoch cap, Fisher pass, one stability-decay
    attempt at the reference's lambda, evaluation of both models: the build's driver + HIP path on the GPU; the same driver +
    the CPU oracle's EWC (oracle/sweep_ref.py) on the host cores, TWICE — at the thread count that won cpu_baseline's probe and at
    half / double of it, in two processes that start AFTER the GPU's sweep (which is timed with nothing else on the host) and run
    beside the GPU's remaining legs (the pair's, the chain's, the kernel-path comparison).  Same task files, start model, batches, head
    initialisation.  `cpu_spread_points` = the largest accuracy difference between the two CPU legs (two summation orders of
    the same fp32 arithmetic); `max_accuracy_gap_points` = the largest between the GPU leg and the first CPU leg.

    `gpu_over_cpu_wall_clock` = `pair.cpu_s / pair.gpu_s`: measured like for like, the ratio to quote.
    `cpu_s_extrapolated`: the GPU sweep's counted image passes priced at the two batch-200 host rates (forward+backward+update,
    forward only) that cpu_baseline measured with the host otherwise idle (`cpu_rates`).  An extrapolation, and says so."""
    import contextlib
    import io
    import shutil
    import subprocess
    import tempfile
    from clsurvey_amd.framework import driver
    from clsurvey_amd.framework.tasks import SyntheticTaskSequence
    from clsurvey_amd.methods import method as M
    root = tempfile.mkdtemp(prefix="clhip_sweep_")
    model = "small_VGG9_cl_128_128"
    dev = "cuda:%d" % dev_index
    res = {"what": "%d-task EWC sweep, %s, %d/%d/%d images of 3x64x64 per task, 20 classes, the reference's defaults "
                   "(LR grid {1e-2,5e-3,1e-3,5e-4,1e-4}, %d-epoch cap, batch 200, lambda 400, drop margin 0.2, --test); torchvision "
                   "initialisation; synthetic 'blobs' tasks with overlapping classes (best possible accuracy %.1f %%)"
                   % ((tasks, model) + tuple(sizes) + (epochs, 100.0 * (SWEEP_DATA["blobs"]["q"] + (1 - SWEEP_DATA["blobs"]["q"]) / 20))),
           "data": {"kind": SWEEP_DATA["kind"], "noise": SWEEP_DATA["noise"], **SWEEP_DATA["blobs"]}}
    quiet = io.StringIO()
    legs = []
    chain_state = None
    try:
        pair = None
        # ---- the full sweep on the GPU (tasks == 0: only the pair, for checks of the pair itself).  FIRST, with nothing else on the host:
        # beside the five torch-CPU processes of the pair / chain legs (~100 busy threads) the same sweep takes 2.7x longer — the
        # host side of the launch path slows down, not the GPU (profiles/r06c_sweep_alone_vs_in_bench.txt: 32.6 s alone, 88 - 99 s
        # beside them, pinning this process to free cores changes nothing) — and BASELINE's 'full-sweep wall-clock' is the GPU path's.
        counts = None
        groot = os.path.join(root, "gpu")
        if tasks > 0:
            ds = SyntheticTaskSequence(os.path.join(groot, "data"), task_count=tasks, classes_per_task=20, sizes=tuple(sizes), hw=64,
                                       name="synthetic_tiny_imagenet", noise=SWEEP_DATA["noise"], kind=SWEEP_DATA["kind"],
                                       blobs=SWEEP_DATA["blobs"], seed=SWEEP_DATA.get("seed", 7))
            t0 = time.perf_counter()
            for i in range(1, tasks + 1):
                ds.get_task_dataset_path(str(i))
            res["task_files_s (not counted)"] = time.perf_counter() - t0
            common = [model, "--num_epochs", str(epochs), "--results_root", groot, "--device", dev]
            if lr_grid:     # (the first task keeps the whole grid: the first-task model's name is made of --lr_grid, net.py:39-53)
                common += ["--lr_grid", lr_grid, "--boot_lr_grid", "1e-2,5e-3,1e-3,5e-4,1e-4"]
            with contextlib.redirect_stdout(quiet), _PassCounter(sizes[0]) as counts:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                driver.main(common + ["--method_name", "SI", "--runmode", "first_task_basemodel_dump"], method=M.parse("SI"), dataset=ds)
                torch.cuda.synchronize()
                res["gpu_first_task_s"] = time.perf_counter() - t0
                try:
                    out = driver.main(common + ["--method_name", "EWC", "--test"], method=M.parse("EWC"), dataset=ds)
                except BaseException as e:
                    # (the reference's own loop ends the same way when every stability-decay attempt of a task diverges:
                    # no best_model.pth.tar for the next task to start from, framework_train.py:143)
                    lines = quiet.getvalue().splitlines()
                    res["gpu_error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
                    res["gpu_log_tail"] = [ln for ln in lines if "ATTEMPT" in ln or "FINETUNE DONE" in ln or "Loss" in ln][-40:]
                    raise
                torch.cuda.synchronize()
                res["gpu_s"] = time.perf_counter() - t0
            r = out["results"]
            res["gpu_image_passes"] = dict(counts)
            res["gpu_phase2_trainings_per_task"] = [len(hf.trace) for hf in out["frameworks"] if hf is not None]
            res["gpu_phase2_attempts"] = [[{"lambda": float(h["lambda"]), "val_acc": float(a), "threshold": float(th)} for h, a, th in hf.trace]
                                          for hf in out["frameworks"] if hf is not None]
            last = torch.load(out["model_paths"][-1], map_location="cpu", weights_only=False)
            om = [v["omega"] for v in getattr(last, "reg_params", {}).values() if isinstance(v, dict) and "omega" in v]
            res["gpu_omega_of_last_model"] = {"max": float(max(float(o.max()) for o in om)), "sum": float(sum(float(o.double().sum()) for o in om))} if om else None
            res["gpu_accepted_lambda_per_task"] = [float(hf.trace[-1][0]["lambda"]) for hf in out["frameworks"] if hf is not None and hf.trace]
            res["gpu_stability"] = sweep_stability(out)
            res["gpu_last_grid"] = [[float(lr), float(a)] for lr, _, a in out["manager"].grid_trace]
            res["gpu_final_accuracies"] = [r[i]["seq_res"][i][-1] for i in sorted(r)]          # task i under the LAST model
            res["gpu_first_accuracies"] = [r[i]["seq_res"][i][0] for i in sorted(r)]           # task i right after training it
            res["gpu_avg_accuracy"] = float(np.mean(res["gpu_final_accuracies"]))
            res["gpu_avg_forgetting"] = float(np.mean([r[i]["seq_forgetting"][i][-1] for i in sorted(r) if r[i]["seq_forgetting"][i]] or [0.0]))
            res["conditioning"] = sweep_conditioning(res["gpu_stability"])
        if cpu_threads:
            # ---- the pair: first-task model on the GPU (untimed), CPU legs started in the background, GPU leg timed
            proot = os.path.join(root, "pair_gpu")
            with contextlib.redirect_stdout(quiet):
                driver.main(_pair_args(dev) + ["--results_root", proot, "--method_name", "SI", "--runmode", "first_task_basemodel_dump"],
                            method=M.parse("SI"))
            other = cpu_threads // 2 if cpu_threads >= 32 else min(2 * cpu_threads, os.cpu_count() or cpu_threads)
            ncpu = os.cpu_count() or 1
            for leg_no, t in enumerate([cpu_threads] + ([other] if other != cpu_threads else [])):
                croot = os.path.join(root, "pair_cpu_t%d" % t)
                # (logical CPUs [0, n/4) and [n/4, n/2): distinct physical cores, on a two-socket host distinct sockets; the
                # upper half are the SMT siblings.  Not pinned on small hosts.)
                pin = leg_no * (ncpu // 4) if ncpu >= 4 * max(cpu_threads, other) else -1
                for sub in ("data", "models", os.path.join("train", "synthetic_tiny_imagenet", "SI")):
                    shutil.copytree(os.path.join(proot, sub), os.path.join(croot, sub))
                env = dict(os.environ, OMP_NUM_THREADS=str(t), MKL_NUM_THREADS=str(t),
                           HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")     # a CPU leg makes no device context
                errf = open(os.path.join(root, "pair_cpu_t%d.stderr" % t), "w+")
                legs.append((t, subprocess.Popen([sys.executable, os.path.abspath(__file__), "--pair-cpu-leg", croot, "--pair-threads", str(t), "--pair-pin", str(pin), "--sweep-blobs",
                                                  ",".join("%g" % SWEEP_DATA["blobs"][k] for k in ("g", "amp", "noise_lr", "q"))],
                                                 stdout=subprocess.PIPE, stderr=errf, env=env, text=True), errf))
            with contextlib.redirect_stdout(quiet), _PassCounter(PAIR["sizes"][0]) as pcounts:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                gout = driver.main(_pair_args(dev) + _pair_fixed() + ["--results_root", proot], method=M.parse("EWC"))
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            pair = {"what": "task 2 of a 2-task sequence (%d/%d/%d images of 3x64x64, 20 classes), from the same first-task model: LR "
                            "grid {%s}, %d-epoch cap, batch %d, Fisher pass, one stability-decay attempt at lambda = %g, both models "
                            "evaluated; same task files, start model, batches and head initialisation on every leg"
                            % (tuple(PAIR["sizes"]) + (PAIR["lr"], PAIR["epochs"], PAIR["batch"], PAIR["lam"])),
                    "gpu": _pair_summary(gout, dt, pcounts)}
            if chain:
                try:
                    # (the pair's legs hold [0, cpu_threads) and [ncpu / 4, ncpu / 4 + other): the chain's legs take what is left of the
                    # second quarter — on a two-socket host the second socket)
                    room = ncpu // 2 - (ncpu // 4 + other)
                    pinned = ncpu >= 4 * max(cpu_threads, other) and room >= 4 * (CHAIN["tasks"] - 1)
                    chain_state = chain_start(root, dev, cpu_threads, ncpu // 4 + other if pinned else None, room if pinned else 0)
                except BaseException as e:     # noqa: BLE001  (the pair and the sweep are reported regardless)
                    res["chain"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:400])}
        # ---- (the full sweep ran first, above) its kernel-path comparison runs beside the CPU legs
        if tasks > 0:
            if forced:
                try:
                    res["forced_paths"] = forced_paths(out, res["gpu_stability"], groot, epochs)
                except BaseException as e:     # noqa: BLE001
                    res["forced_paths"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:400])}
            res["chance_accuracy"] = 100.0 / 20
            res["best_possible_accuracy"] = 100.0 * (SWEEP_DATA["blobs"]["q"] + (1 - SWEEP_DATA["blobs"]["q"]) / 20)
        # ---- collect the CPU legs
        if chain_state is not None:
            try:
                res["chain"] = chain_collect(chain_state)
            except BaseException as e:     # noqa: BLE001
                res["chain"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:400])}
        if pair is not None:
            cpu = []
            for t, proc, errf in legs:
                try:
                    so, _ = proc.communicate(timeout=900)
                except subprocess.TimeoutExpired:
                    proc.kill()
                    so, _ = proc.communicate()
                errf.seek(0)
                se = errf.read()
                errf.close()
                line = [ln for ln in so.splitlines() if ln.startswith("{")]
                if proc.returncode != 0 or not line:
                    raise RuntimeError("pair CPU leg (%d threads) failed:\n%s" % (t, se[-2000:]))
                cpu.append(json.loads(line[-1]))
            legs = []
            pair["cpu"] = cpu[0]
            if len(cpu) > 1:
                pair["cpu_other_threads"] = cpu[1]

            def gap(a, b):
                return max(abs(x - y) for i in a["accuracies"] for x, y in zip(a["accuracies"][i], b["accuracies"][i]))
            pair["max_accuracy_gap_points"] = gap(pair["gpu"], cpu[0])
            pair["cpu_spread_points"] = gap(cpu[0], cpu[1]) if len(cpu) > 1 else None
            pair["agree"] = pair["max_accuracy_gap_points"] <= max(3.0, pair["cpu_spread_points"] or 0.0)
            pair["gpu_s"], pair["cpu_s"] = pair["gpu"]["seconds"], cpu[0]["seconds"]
            pair["cpu_concurrency"] = "the two CPU legs (%s threads) ran side by side%s, after the GPU sweep, beside the GPU's pair / chain / kernel-path legs (%d logical cores)" % (
                " / ".join(str(c["threads"]) for c in cpu),
                (", next to the chain's %d CPU legs (%s threads each, logical CPUs from %s)" % (
                    res["chain"].get("tasks_compared", 0), res["chain"].get("cpu_threads_per_leg"), res["chain"].get("cpu_legs_pinned_from_logical_cpu"))
                 if isinstance(res.get("chain"), dict) and "per_task" in res["chain"] else ""), os.cpu_count() or 0)
            pair["gpu_over_cpu_wall_clock"] = pair["cpu_s"] / pair["gpu_s"]
            res["pair"] = pair
            res["pair_cpu_rates_images_per_s"] = cpu[0]["rates_images_per_s"]
            # the headline ratio is the MEASURED like-for-like one (same work on both sides)
            res["gpu_over_cpu_wall_clock"] = pair["gpu_over_cpu_wall_clock"]
            res["gpu_over_cpu_wall_clock_how"] = "pair: cpu_s / gpu_s of the same bounded task, both measured in this run"
            if counts is not None and cpu_rates:
                # The sweep runs at batch 200; the pair's legs run at batch 50, two of them side by side with the chain's legs, so their
                # rates under-state the host.  Price the sweep's passes at the batch-200 rates cpu_baseline measured ALONE on the
                # host (best thread count of its probe) instead.
                res["cpu_rates_images_per_s"] = dict(cpu_rates)
                res["cpu_s_extrapolated"] = counts["train"] / cpu_rates["forward_backward_update"] + counts["eval"] / cpu_rates["forward_only"]
                res["cpu_s_extrapolated_how"] = ("GPU sweep's image passes (%d forward+backward, %d forward-only) at the batch-200 host rates "
                                                 "cpu_baseline measured with nothing else running on this box; an extrapolation, not run"
                                                 % (counts["train"], counts["eval"]))
                res["gpu_over_cpu_extrapolated"] = res["cpu_s_extrapolated"] / res["gpu_s"]
    except BaseException as e:
        if "gpu_error" not in res:
            raise
        res["error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
    finally:
        for _, proc, errf in legs:
            proc.kill()
            errf.close()
        for proc, errf in (chain_state or {}).get("procs", []):
            proc.kill()
            errf.close()
        shutil.rmtree(root, ignore_errors=True)
    return res


def sharded_sweep(dev_index, world, epochs=8, sizes=(2000, 500, 500), batch=50):
    """SURVEY 8(e) as the build runs it — `driver.main(... --shard)` on a bounded 2-task EWC sequence of SWEEP_DATA tasks, the SAME
    work at every world size (so that 1 -> 8 GPUs is a strong-scaling curve of the framework itself; at world 1 --shard
    changes nothing and this is the sequential driver): phase-1 grid nodes spread over the ranks (all_gather of accuracies,
    winner's model files broadcast from the rank that trained it), speculative stability decay (one attempt per rank in
    flight, accepted attempt's files broadcast), evaluation pairs spread over the ranks (all_gather).  Every rank calls this;
    returns seconds, fill factor, what THIS rank did per stage and its collective traffic — models and metrics only,
    nothing on the per-batch path."""
    import contextlib
    import io
    import shutil
    import tempfile
    from clsurvey_amd.framework import driver, shard
    from clsurvey_amd.methods import method as M
    root = tempfile.mkdtemp(prefix="clhip_shard_")
    b = SWEEP_DATA["blobs"]
    spec = "2,20,%d,%d,%d,64,%g,%s,%g,%g,%g,%g" % (tuple(sizes) + (SWEEP_DATA["noise"], SWEEP_DATA["kind"], b["g"], b["amp"], b["noise_lr"], b["q"]))
    common = ["small_VGG9_cl_128_128", "--num_epochs", str(epochs), "--batch_size", str(batch), "--synthetic", spec,
              "--device", "cuda:%d" % dev_index, "--results_root", root, "--shard"]
    before = dict(shard.STATS)
    quiet = io.StringIO()
    try:
        with contextlib.redirect_stdout(quiet):
            shard.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            driver.main(common + ["--method_name", "SI", "--runmode", "first_task_basemodel_dump"], method=M.parse("SI"))
            t1 = time.perf_counter()
            out = driver.main(common + ["--method_name", "EWC", "--test"], method=M.parse("EWC"))
            torch.cuda.synchronize()
            shard.barrier()
            dt = time.perf_counter() - t0
        hf = out["frameworks"][-1]
        d = {k: shard.STATS[k] - before[k] for k in shard.STATS}
        d["grid_nodes_total"] = 10                                       # two 5-LR grids (first task, task 2): every node's result is used
        busy = d["grid_busy_s"] + d["decay_busy_s"] + d["eval_busy_s"]
        return {"what": "driver.main --shard: SI first-task grid + EWC task 2 (5-LR grid, stability decay from lambda 400, "
                        "evaluation), %d/%d/%d images of 3x64x64 per task, batch %d, %d-epoch cap — the same work at every world size"
                        % (tuple(sizes) + (batch, epochs)),
                "world": world, "seconds": dt, "first_task_seconds": t1 - t0,
                "grid_nodes_per_task": 5, "fill_factor_grid": shard.fill_factor(5, world),
                "phase2_trainings_task2": len(hf.trace), "accepted_lambda_task2": float(hf.trace[-1][0]["lambda"]) if hf.trace else None,
                # speculation accounting (all ranks together, the same numbers on every rank): phase-2 trainings that ran, and those
                # the sequential rule would have run too (attempts up to the accepted one); world 1 runs nothing speculatively
                "phase2_trainings_all_ranks": d["decay_trainings_group"] if world > 1 else d["decay_attempts"],
                "phase2_trainings_useful": d["decay_trainings_useful"] if world > 1 else d["decay_attempts"],
                "useful_fraction": ((d["grid_nodes_total"] + (d["decay_trainings_useful"] if world > 1 else d["decay_attempts"]))
                                    / max(d["grid_nodes_total"] + (d["decay_trainings_group"] if world > 1 else d["decay_attempts"]), 1)),
                "accuracies": {i: r["seq_res"][i] for i, r in sorted(out["results"].items())},
                "this_rank": {"grid_nodes": d["grid_nodes"], "grid_busy_s": d["grid_busy_s"],
                              "decay_attempts": d["decay_attempts"], "decay_busy_s": d["decay_busy_s"],
                              "eval_pairs": d["eval_pairs"], "eval_busy_s": d["eval_busy_s"],
                              "busy_fraction": busy / dt if dt > 0 else 0.0},
                "collectives_this_rank": {k: d[k] for k in ("broadcast_calls", "broadcast_bytes", "broadcast_s", "all_gather_calls",
                                                            "all_gather_bytes", "all_gather_s", "all_reduce_calls", "all_reduce_s")}}
    finally:
        shutil.rmtree(root, ignore_errors=True)


LINE_LIMIT = 6000        # bytes: the driver keeps the last 8 KB of stdout and parses its LAST line


def _round_floats(o, digits=6):
    """Floats to `digits` significant digits (the line is a record, not an archive: full precision lives in the details file)."""
    if isinstance(o, float):
        return float("%.*g" % (digits, o)) if o == o and abs(o) != float("inf") else None
    if isinstance(o, dict):
        return {k: _round_floats(v, digits) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_round_floats(v, digits) for v in o]
    return o


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(out, details_path=None, limit=LINE_LIMIT):
    """The ONE stdout line: the contract's keys + `roofline` + `cpu_baseline` + a few scalars of the other objects, at most
    `limit` bytes.  Everything else (per-layer table, the other BASELINE configs, the sweep's body) goes to the details file
    and to stderr.  Optional parts are dropped, longest first, if the line would still exceed the limit; the contract's keys
    never are."""
    c = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data"))
    c["vs_baseline"] = out.get("vs_baseline")
    c["config"] = _pick(out.get("config", {}), ("workload", "images_per_step", "batch", "parallelism", "algorithmic_gflop_per_step",
                                                "step_algorithmic_tflops", "step_mfma_issued_frac", "gpu_over_cpu"))
    if "roofline" in out:
        r = out["roofline"]
        c["roofline"] = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_gflop_per_launch",
                                  "algorithmic_bytes_per_launch", "avg_launch_us", "avg_launch_how", "path", "peak_how",
                                  "mfma_issued_frac", "mfma_busy_pmc", "share_of_conv_launch_time", "launches_per_pass", "longest_launch",
                                  "by_share"))
        c["roofline"].setdefault("traffic", r.get("traffic"))
    if "cpu_baseline" in out:
        c["cpu_baseline"] = _pick(out["cpu_baseline"], ("value", "unit", "cores", "kind", "sample", "host_logical_cores", "host_cpu_model",
                                                        "forward_only_images_per_s", "thread_probe_images_per_s"))
    optional = {}
    if "sweep_s" in out:
        optional["sweep_s"] = out["sweep_s"]
    if isinstance(out.get("grid"), dict):
        optional["grid"] = _pick(out["grid"], ("nodes", "iterations", "winner_rank", "winner_lr", "fill_factor", "collectives_in_timed_region",
                                                     "per_rank_ms_per_step", "collective_seconds_max_over_ranks", "host_marks_s_rank0", "backend", "visible_devices"))
    ss = out.get("sharded_sweep")
    if isinstance(ss, dict):
        optional["sharded_sweep"] = _pick(ss, ("error", "world", "seconds", "first_task_seconds", "methods", "fill_factor_grid", "fill_factor",
                                               "useful_fraction", "phase2_trainings_task2", "accepted_lambda_task2"))
        if "this_rank" in ss:
            optional["sharded_sweep"]["busy_fraction_rank0"] = ss["this_rank"].get("busy_fraction")
    cfgs = out.get("configs")
    if isinstance(cfgs, dict):
        optional["configs_ms_per_step"] = {k: v["ms_per_step"] for k, v in cfgs.items() if isinstance(v, dict) and "ms_per_step" in v}
        cb = cfgs.get("conv_backward")
        if isinstance(cb, dict):
            optional["conv_backward_mfma_issued_frac"] = {k: v.get("all_backward_mfma_issued_frac") for k, v in cb.items() if isinstance(v, dict)}
    if details_path:
        optional["details"] = details_path
    c.update(optional)
    c = _round_floats(c)
    line = json.dumps(c, separators=(",", ":"))
    for k in sorted(optional, key=lambda k: -len(json.dumps(c.get(k)))):
        if len(line) <= limit:
            break
        del c[k]
        line = json.dumps(c, separators=(",", ":"))
    if len(line) > limit:                      # last resort: the free-text fields
        for path in (("roofline", "avg_launch_how"), ("roofline", "peak_how"), ("cpu_baseline", "host_cpu_model"), ("cpu_baseline", "sample")):
            if len(line) <= limit:
                break
            if path[0] in c and path[1] in c[path[0]]:
                c[path[0]][path[1]] = str(c[path[0]][path[1]])[:80]
                line = json.dumps(c, separators=(",", ":"))
    return line


def emit(out):
    """Details to gpurun_out/bench_details.json (when writable) and to stderr; the compact line LAST and ALONE on stdout."""
    path = None
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "bench_details.json")
        with open(path, "w") as f:
            json.dump(out, f)
        path = os.path.relpath(path, ROOT)
    except OSError:
        path = None
    sys.stderr.write("bench-details: " + json.dumps(out) + "\n")
    sys.stderr.flush()
    sys.stdout.flush()
    print(compact_line(out, path), flush=True)


def host_cpu():
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"logical_cores": os.cpu_count() or 1, "model": model}


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
    probe = {}
    for cand in sorted({min(ncpu, c) for c in (16, 32, 64, 128, 256)}):
        torch.set_num_threads(cand)
        step(best is None)           # warm-up at this thread count (first call creates the buffers)
        t0 = time.perf_counter()
        step(False)
        t = time.perf_counter() - t0
        probe[cand] = t
        if best is None or t < best[1]:
            best = (cand, t)
        # (past the knee larger counts only get slower, and the probe must stay short: at 256 threads one warm-up + one timed step took
        # 90 s — 9 images/s, profiles/r06b_final_bench_line.json — behind 182 images/s at 128 and 506 at 32)
        if t > 2.0 * best[1]:
            break
    cores = best[0]
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    for _ in range(steps):
        step(False)
    dt = time.perf_counter() - t0
    # forward only (evaluation / validation passes of a sweep), same batch, same threads, nothing else on the host
    with torch.no_grad():
        vgg_ref.forward(params, SMALL, x)
        t0 = time.perf_counter()
        nf = max(2, steps // 2)
        for _ in range(nf):
            vgg_ref.forward(params, SMALL, x)
        dtf = time.perf_counter() - t0
    cpu = host_cpu()
    return dict(value=2 * batch * steps / dt, unit="images/s", cores=cores, kind="port", host_logical_cores=cpu["logical_cores"],
                host_cpu_model=cpu["model"], forward_only_images_per_s=batch * nf / dtf,
                thread_probe_images_per_s={str(k): 2 * batch / v for k, v in sorted(probe.items())},
                sample="%d steps (EWC train batch + Fisher batch, N=%d) of the torch-CPU oracle, %d threads (fastest of %s "
                       "on this host's %d logical cores: %s images/s in a one-step probe; one process, as the reference runs), %.1f s"
                       % (steps, batch, cores, " / ".join(str(k) for k in sorted(probe)), cpu["logical_cores"],
                          " / ".join("%.0f" % (2 * batch / probe[k]) for k in sorted(probe)), dt))


def main():
    args = parse()
    if args.sweep_blobs:
        SWEEP_DATA["blobs"] = dict(zip(("g", "amp", "noise_lr", "q"), (float(v) for v in args.sweep_blobs.split(","))))
    if args.pair_cpu_leg:                      # a CPU leg of the sweep pair, in its own process (no GPU work)
        print(json.dumps(pair_cpu_leg(args.pair_cpu_leg, args.pair_threads, args.pair_pin if args.pair_pin >= 0 else None)), flush=True)
        return
    if args.chain_cpu_leg:
        print(json.dumps(chain_cpu_leg(args.chain_cpu_leg, args.chain_job, args.pair_threads, args.pair_pin if args.pair_pin >= 0 else None)), flush=True)
        return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    local_rank %= torch.cuda.device_count()          # (dry runs of the N > 1 path on a 1-GPU box: CLHIP_BENCH_BACKEND=gloo)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    backend = os.environ.get("CLHIP_BENCH_BACKEND", "nccl")
    # CLHIP_BENCH_FORCE_DIST=1: the N > 1 code path with a communicator of ONE rank (dry run of the RCCL calls on a 1-GPU box: every
    # collective of this file and of framework/shard.py is issued over the real backend; tools/experiments/r06c_rccl1.sh)
    force_dist = world == 1 and os.environ.get("CLHIP_BENCH_FORCE_DIST") == "1"
    multi = world > 1 or force_dist
    if force_dist:
        os.environ["CLHIP_SHARD_FORCE_COLLECTIVES"] = "1"
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def bcast(t, src):
        """RCCL broadcast of a device tensor (gloo dry runs stage through the host)."""
        if backend == "nccl":
            dist.broadcast(t, src=src)
        else:
            h = t.cpu()
            dist.broadcast(h, src=src)
            t.copy_(h)

    if args.forced_leg:
        print(json.dumps(forced_leg(args.forced_leg, "cuda:%d" % local_rank)), flush=True)
        return
    if args.sweep_only:
        print(json.dumps(full_sweep(local_rank, cpu_threads=0 if args.no_cpu_baseline else 16, tasks=args.sweep_tasks,
                                    epochs=args.sweep_epochs)), flush=True)
        return
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

    # N > 1: rank r is grid node r of the phase-1 LR grid (lr_grid_train.py:51-151) — its own LR, same start model
    LR_GRID = [1e-2, 5e-3, 1e-3, 5e-4, 1e-4]                                # framework/main.py:61
    lr = LR_GRID[rank % len(LR_GRID)] if world > 1 else 1e-3

    def step(i, first=False):
        idx = perm[(i % nb) * N:(i % nb + 1) * N]
        x = data_x.index_select(0, idx)
        y = data_y.index_select(0, idx)
        eng.loss_step(x, y, "ce_mean", True, stats)                       # train_EWC.py:181-187
        ops.reg_sgd_step(A.theta, A.grad, omega, init_val, buf, 400.0, lr, 0.9, 0.0, first)     # :189
        j = (i % nb) * N
        eng.loss_step(data_x[j:j + N], data_y[j:j + N], "ce_sum", True)   # main_EWC.py:147-149
        ops.fisher_accum(fisher, A.grad, 8000.0)                          # :155

    step(0, True)
    for i in range(args.warmup):
        step(i + 1)
    # The dominant launch of a pass (longest single launch, found by timing every conv launch of the plan back to back) is
    # then timed INSIDE the timed steps: clhip_net_probe_kind keeps a ring of HIP event pairs around that layer's forward /
    # backward-data / weight-gradient launch on the stream it is issued on, read after the loop
    rows, dom = None, None
    if rank == 0:
        rows = time_kernels(eng, data_x[:N].contiguous(), N, args.kernel_iters)
        # dominant = the kernel INSTANCE with the largest share of a pass (sum over its launches), not the longest single
        # launch; its longest launch is the one probed in situ
        # (the forward and the backward-data launch of a layer are two template instances of ONE kernel — rocprofv3's stats list them under
        # one name: grouped by that name, mode / un-pool arguments dropped)
        import re

        def family(inst):
            return re.sub(r">, [01], (true|false)> ", ">, mode, unpool> ", inst) if inst.startswith("bs_conv_") else inst
        by_inst = {}
        for r in rows:
            by_inst.setdefault(family(r["instance"]), []).append(r)
        dom_rows = max(by_inst.values(), key=lambda rs: sum(r["sec"] for r in rs))
        dom = max(dom_rows, key=lambda r: r["sec"])
        longest = max(rows, key=lambda r: r["sec"])
        eng.probe(dom["li"], dom["kind"])
    torch.cuda.synchronize()
    if dist:
        # the collectives of the timed region once, untimed, on buffers of the same shapes: the first broadcast / all_gather of a
        # process pays the backend's one-off set-up (streams, events, kernel load: 8 - 28 ms measured on a one-rank RCCL communicator,
        # profiles/r06c_rccl_one_rank.txt) — as much as the 20 steps they bracket
        # ... and so does the first launch of every torch kernel the exchange uses (the float64 division / stack / scalar upload that build
        # the metric row: 18 - 52 ms on an idle GPU the first time, tools/experiments/r06c_rccl4.sh) — so the exchange's own statements run here
        wb = torch.empty_like(A.theta)
        bcast(wb, 0)
        wm = torch.stack([stats[1] / max(float(args.steps * N), 1.0), torch.tensor(float(rank), dtype=torch.float64, device=dev)])
        if backend != "nccl":
            wm = wm.cpu()
        wt = [torch.zeros_like(wm) for _ in range(world)]
        dist.all_gather(wt, wm)
        int(max(wt, key=lambda t: (float(t[0]), -float(t[1])))[1])
        bcast(wb, world - 1)
        del wb, wm, wt
        torch.cuda.synchronize()
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    grid = None
    coll_ev = []          # (name, start event, end event) on the current stream around each collective of the timed region

    def timed_collective(name, fn):
        if backend == "nccl":
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            coll_ev.append((name, a, b))
        else:
            t = time.perf_counter()
            fn()
            coll_ev.append((name, time.perf_counter() - t, None))
    marks = []            # (what has been ENQUEUED / finished on the host, seconds since t0): where a rank's time goes when N > 1
    if dist:
        timed_collective("broadcast_start_arena", lambda: bcast(A.theta, 0))   # every node starts from the previous task's model
        stats.zero_()
        marks.append(("start_broadcast_enqueued", time.perf_counter() - t0))
    for i in range(args.steps):
        step(i)
    marks.append(("steps_enqueued", time.perf_counter() - t0))
    if dist:
        # metrics of the N nodes to every rank, the winner's parameter arena back to every rank
        mine = torch.stack([stats[1] / max(float(args.steps * N), 1.0), torch.tensor(float(rank), dtype=torch.float64, device=dev)])
        marks.append(("metrics_tensor_built", time.perf_counter() - t0))
        if backend != "nccl":
            mine = mine.cpu()
        table = [torch.zeros_like(mine) for _ in range(world)]
        marks.append(("table_allocated", time.perf_counter() - t0))
        timed_collective("all_gather_node_metrics", lambda: dist.all_gather(table, mine))
        marks.append(("all_gather_enqueued", time.perf_counter() - t0))
        winner = int(max(table, key=lambda t: (float(t[0]), -float(t[1])))[1])
        marks.append(("metrics_on_host (steps finished)", time.perf_counter() - t0))
        timed_collective("broadcast_winner_arena", lambda: bcast(A.theta, winner))
        marks.append(("winner_broadcast_enqueued", time.perf_counter() - t0))
        grid = {"nodes": world, "lr_grid": LR_GRID, "iterations": -(-world // len(LR_GRID)), "winner_rank": winner,
                "winner_lr": LR_GRID[winner % len(LR_GRID)],
                "collectives_in_timed_region": ["broadcast(start arena %.1f MB)" % (A.numel * 4 / 1e6),
                                                "all_gather(node metrics)", "broadcast(winner arena)"],
                "fill_factor": shard_fill(len(LR_GRID), world)}
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0            # this rank's own time (before it waits for the others)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # per-rank record (outside the timed region): every rank's own ms per step and the seconds its collectives took on its stream,
        # so that a weak-scaling value that is not N x the single-GPU one can be read from the line (slow rank vs. slow collective)
        mine_s = [dt_local] + [(a.elapsed_time(b) * 1e-3 if b is not None else float(a)) for _, a, b in coll_ev]
        tt = torch.tensor(mine_s, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        allr = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allr, tt)
        per_rank = [[float(v) for v in r.cpu()] for r in allr]
        grid["per_rank_ms_per_step"] = [r[0] / args.steps * 1e3 for r in per_rank]
        grid["collective_seconds_max_over_ranks"] = {name: max(r[1 + j] for r in per_rank) for j, (name, _, _) in enumerate(coll_ev)}
        grid["collective_seconds_rank0"] = {name: per_rank[0][1 + j] for j, (name, _, _) in enumerate(coll_ev)}
        grid["backend"] = "rccl" if backend == "nccl" else backend
        grid["host_marks_s_rank0"] = {k: v for k, v in marks}
        grid["visible_devices"] = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES"))
    if not torch.isfinite(A.theta).all():
        raise SystemExit("non-finite parameters after the timed run")
    probed_us, probed_n = eng.probe_read() if dom is not None else (0.0, 0)
    if dom is not None:
        eng.probe(None)

    fwd_fl, step_fl = algorithmic_flops_per_image(SMALL, (128, 128), 20, 64)
    step_iss = step_flops_per_image(eng, 64)[1]          # matrix-pipe seconds of the same step per image, at the pipes' peaks
    imgs = 2 * N * args.steps * world
    out = {
        "metric": "images/sec (train+importance pass), EWC small_VGG9 Tiny-ImageNet task batch",
        "value": imgs / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": "EWC small_VGG9_cl_128_128, Tiny-ImageNet shapes (3x64x64, 20 classes), "
                               "batch 200: train step + Fisher step (BASELINE configs[1])",
                   "images_per_step": 2 * N, "batch": N,
                   "parallelism": ("1 GPU" if not multi else "1 GPU, communicator of one rank (dry run of the N > 1 path)" if force_dist else
                                   "%d grid nodes of the phase-1 LR grid, one per GPU (RCCL: start-model broadcast, metric "
                                   "all_gather, winner broadcast)" % world),
                   "algorithmic_gflop_per_step": 2 * N * step_fl / 1e9,
                   "step_algorithmic_tflops": 2 * N * step_fl * args.steps / dt / 1e12,
                   "step_algorithmic_over_f32_mfma_peak": 2 * N * step_fl * args.steps / dt / 1e12 / PEAK_F32_MFMA_TFLOPS,
                   "step_mfma_issued_frac": 2 * N * step_iss * args.steps / dt},
    }
    if grid is not None:
        out["grid"] = grid
    if not args.no_sweep:
        # second number of the line at EVERY N (1 included: the anchor of the curve): the framework itself through
        # `driver --shard` on a bounded task sequence, the same work whatever the world size (every rank takes part)
        try:
            out["sharded_sweep"] = sharded_sweep(local_rank, world)
        except BaseException as e:       # the headline line is printed regardless (stage failures are collective: shard.all_ok)
            out["sharded_sweep"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    if rank == 0:
        agg = {}
        for r in rows:
            a = agg.setdefault(r["kernel"], dict(flops=0.0, sec=0.0, launches=0))
            a["flops"] += r["flops"]; a["sec"] += r["sec"]; a["launches"] += 1
        in_situ = probed_n > 0
        dom_sec = probed_us * 1e-6 if in_situ else dom["sec"]
        ach = dom["flops"] / dom_sec / 1e12
        # `achieved` prices the ALGORITHMIC flops of the convolution.  `peak` is the ceiling of that figure on the path the
        # launch takes: the f32-MFMA peak for a direct f32 kernel; x 36/16 for Winograd F(2x2,3x3) (16 of the direct form's 36
        # multiplies are issued); the dense bf16-MFMA peak / 6 for the bf16-split kernel (six bf16 products per fp32 multiply).
        # `frac` = achieved / peak is then a fraction of what the matrix pipe can do (<= 1; it equals mfma_issued_frac).
        peak = dom["flops"] / dom["pipe_sec"] / 1e12
        peak_how = {"direct": "fp32-MFMA peak %.1f TFLOP/s (MI355X_MICROARCH.md)" % PEAK_F32_MFMA_TFLOPS,
                    "wino": "fp32-MFMA peak %.1f TFLOP/s x 36/16 (F(2x2,3x3): 16 of the direct form's 36 multiplies are issued)" % PEAK_F32_MFMA_TFLOPS,
                    "bs": "dense bf16-MFMA peak %.0f TFLOP/s / 6 (fp32 operands split into 3 bf16 pieces: 6 bf16 products per multiply, "
                          "fp32 accumulation; error = an fp32 chain's, profiles/r05_bf16_split_dot.txt)" % PEAK_BF16_MFMA_TFLOPS}[dom["path"]]
        out["roofline"] = {"bound": "mfma", "kernel": "%s [%s, layer %s, N=%d]" % (dom["instance"], dom["kernel"], dom["layer"], N),
                           "achieved": ach, "peak": peak,
                           "unit": "TFLOP/s", "frac": ach / peak,
                           "peak_how": peak_how, "path": dom["path"],
                           "algorithmic_over_f32_mfma_peak": ach / PEAK_F32_MFMA_TFLOPS,
                           "traffic": measured_traffic(dom["kernel"], dom["layer"], N, dom["instance"]),
                           "algorithmic_gflop_per_launch": dom["flops"] / 1e9,
                           "algorithmic_bytes_per_launch": dom["alg_bytes"],
                           "avg_launch_us": dom_sec * 1e6,
                           "avg_launch_how": ("HIP events around this launch inside the timed steps (clhip_net_probe_kind), last %d passes" % probed_n)
                                             if in_situ else "HIP events around %d back-to-back launches after the timed steps" % args.kernel_iters,
                           "back_to_back_us": dom["sec"] * 1e6,
                           "winograd": dom["path"] == "wino",
                           "mfma_issued_frac": dom["pipe_sec"] / dom_sec,
                           "mfma_busy_pmc": mfma_busy_pmc("small_VGG9", dom["instance"]),
                           "chosen_by": "largest share of the conv launches of a pass, summed over the launches of one kernel instance",
                           "share_of_conv_launch_time": sum(r["sec"] for r in dom_rows) / sum(r["sec"] for r in rows),
                           "launches_per_pass": len(dom_rows),
                           "longest_launch": {"kernel": "%s [%s, layer %s]" % (longest["instance"], longest["kernel"], longest["layer"]),
                                              "us": longest["sec"] * 1e6, "frac": longest["pipe_sec"] / longest["sec"],
                                              "share": longest["sec"] / sum(r["sec"] for r in rows)},
                           "by_share": [{"kernel": inst, "launches": len(rs), "share": sum(r["sec"] for r in rs) / sum(r["sec"] for r in rows),
                                         "frac": sum(r["pipe_sec"] for r in rs) / sum(r["sec"] for r in rs),
                                         "busy_pmc": _weighted_busy(rs)}
                                        for inst, rs in sorted(by_inst.items(), key=lambda kv: -sum(r["sec"] for r in kv[1]))[:3]],
                           "per_kernel": {k: {"tflops": v["flops"] / v["sec"] / 1e12, "us_per_step_pass": v["sec"] * 1e6}
                                          for k, v in agg.items()},
                           "per_layer": [{"kernel": r["kernel"], "layer": r["layer"], "instance": r["instance"], "path": r["path"],
                                          "us": r["sec"] * 1e6, "algorithmic_tflops": r["flops"] / r["sec"] / 1e12,
                                          "algorithmic_over_f32_mfma_peak": r["flops"] / r["sec"] / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                          "mfma_issued_frac": r["pipe_sec"] / r["sec"],
                                          **({"one_grid_with_the_layers_other_backward_launch": True,
                                              "one_grid_us_both_launches_with_transform_and_reduction": r["one_grid_us_both_launches"]}
                                             if r.get("one_grid") else {})} for r in rows]}
        if not multi and not args.no_configs:
            del eng
            torch.cuda.empty_cache()
            out["configs"] = extra_configs(dev, N, args.config_steps)
            torch.cuda.empty_cache()
        if not multi and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, args.cpu_steps)
            out["config"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        if not multi and not args.no_sweep:
            try:
                cb = out.get("cpu_baseline")
                out["sweep"] = full_sweep(local_rank, cpu_threads=0 if args.no_cpu_baseline else cb["cores"],
                                          tasks=args.sweep_tasks, epochs=args.sweep_epochs,
                                          cpu_rates=None if cb is None else {"forward_backward_update": cb["value"],
                                                                             "forward_only": cb["forward_only_images_per_s"]})
            except BaseException as e:            # (incl. the SystemExit of a failed training) the headline line is printed regardless
                import traceback
                out["sweep"] = {"error": "%s: %s" % (type(e).__name__, e), "traceback_tail": traceback.format_exc()[-1500:]}
            out["sweep_s"] = {"gpu": out["sweep"].get("gpu_s"), "cpu_extrapolated": out["sweep"].get("cpu_s_extrapolated"),
                              "gpu_over_cpu_measured_pair": out["sweep"].get("gpu_over_cpu_wall_clock"),
                              "gpu_over_cpu_extrapolated": out["sweep"].get("gpu_over_cpu_extrapolated"),
                              "pair_gpu": out["sweep"].get("pair", {}).get("gpu_s"), "pair_cpu": out["sweep"].get("pair", {}).get("cpu_s"),
                              "pair_max_accuracy_gap_points": out["sweep"].get("pair", {}).get("max_accuracy_gap_points"),
                              "pair_cpu_spread_points": out["sweep"].get("pair", {}).get("cpu_spread_points")}
            fp, cond = out["sweep"].get("forced_paths") or {}, out["sweep"].get("conditioning") or {}
            if out["sweep"].get("error") or out["sweep"].get("gpu_error"):       # (a sequence that ends without a model: profiles/r06b_sweep_redraw.txt)
                out["sweep_s"]["error"] = str(out["sweep"].get("gpu_error") or out["sweep"].get("error"))[:160]
            ch = out["sweep"].get("chain") or {}
            out["sweep_s"]["cpu_vs_hip_chain"] = ({"tasks": ch.get("tasks_compared"), "max_gap_points": ch.get("max_gap_points"),
                                                   "max_omega_sum_rel_gap": ch.get("max_omega_sum_rel_gap"),
                                                   "cpu_over_gpu_seconds": ch.get("cpu_over_gpu_seconds")}
                                                  if "per_task" in ch else ch.get("error"))
            out["sweep_s"]["avg_accuracy"] = out["sweep"].get("gpu_avg_accuracy")
            out["sweep_s"]["avg_forgetting"] = out["sweep"].get("gpu_avg_forgetting")
            # the three kernel paths task by task (teacher-forced on the sweep that just ran), and where the free run's stability-
            # decay decisions sit against the heavy-ball limit (sweep_conditioning)
            out["sweep_s"]["kernel_paths_max_gap_points"] = fp.get("max_gap_all_tasks_points", fp.get("error"))
            out["sweep_s"]["kernel_paths_max_omega_sum_rel_spread"] = max([e["omega_sum_rel_spread"] for e in fp.get("per_task", [])] or [None]) if "per_task" in fp else None
            out["sweep_s"]["decisions"] = _pick(cond, ("attempts", "rejected", "below_limit_rejected", "rejected_all_above_limit", "accepted_above_limit", "min_margin")) if cond else None
        emit(out)
    if dist:
        dist.barrier()          # rank 0 was still timing kernels: tear the communicator down together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
