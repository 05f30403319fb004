#!/usr/bin/env python
"""Per-layer HIP-event timing of the conv kernels for a VGG config (tuning aid).
usage: [CLHIP_LIB=...] python tools/conv_bench.py [small|base|wide] [N] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clsurvey_amd import models, ops  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "small"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cfg = models.CFG[name + "_VGG9"]
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()


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


c, hw = 3, 64
tot = {"fwd": 0.0, "bwd_data": 0.0, "bwd_weight": 0.0}
flt = {"fwd": 0.0, "bwd_data": 0.0, "bwd_weight": 0.0}
print("lib:", os.environ.get("CLHIP_LIB", "default"))
i = 0
while i < len(cfg):
    k = cfg[i]
    x = torch.randn(N, c, hw, hw, device=dev)
    w = torch.randn(k, c, 3, 3, device=dev) * 0.05
    b = torch.zeros(k, device=dev)
    dy = torch.randn(N, k, hw, hw, device=dev)
    fl = 2.0 * 9 * c * k * hw * hw * N
    t = {"fwd": timed(lambda: ops.conv3x3_fwd(x, w, b, True)),
         "bwd_weight": timed(lambda: ops.conv3x3_bwd_weight(x, dy))}
    if c > 3:
        t["bwd_data"] = timed(lambda: ops.conv3x3_bwd_data(dy, w, x))
    pooled = i + 1 < len(cfg) and cfg[i + 1] == "M"
    if pooled:
        t["fwd+pool"] = timed(lambda: ops.conv3x3_relu_pool_fwd(x, w, b))
        if c == 3:
            yp, idx = ops.conv3x3_relu_pool_fwd(x, w, b)
            dyp = torch.randn_like(yp)
            t["wgrad_unpool"] = timed(lambda: ops.conv3x3_bwd_weight_unpool(x, dyp, idx))
    line = "%4dx%-4d@%-3d" % (c, k, hw)
    for kk in ("fwd", "bwd_data", "bwd_weight"):
        if kk in t:
            tot[kk] += t[kk]
            flt[kk] += fl
            line += "  %s %7.1f us %6.1f TF" % (kk, t[kk] * 1e6, fl / t[kk] / 1e12)
    for kk in ("fwd+pool", "wgrad_unpool"):
        if kk in t:
            line += "  %s %7.1f us %6.1f TF" % (kk, t[kk] * 1e6, fl / t[kk] / 1e12)
    print(line)
    c = k
    i += 1
    if i < len(cfg) and cfg[i] == "M":
        hw //= 2
        i += 1
for kk in tot:
    print("TOTAL %-10s %8.1f us  %6.1f TF" % (kk, tot[kk] * 1e6, flt[kk] / tot[kk] / 1e12))
print("ALL %.1f us" % (sum(tot.values()) * 1e6))
