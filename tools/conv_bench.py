#!/usr/bin/env python
"""Per-layer HIP-event timing of the conv launches the plan executor issues for one pass (Winograd or direct, as the plan
chose), for a VGG9 width.  usage: conv_bench.py {small|base|wide} [N] [iters]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from clsurvey_amd import models, net  # noqa: E402
which = sys.argv[1] if len(sys.argv) > 1 else "small"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
it = int(sys.argv[3]) if len(sys.argv) > 3 else 20
name = {"small": "small_VGG9_cl_128_128", "base": "base_VGG9_cl_512_512", "wide": "wide_VGG9_cl_512_512"}[which]
eng = net.NetEngine(models.parse_model_name(name, (64, 64), 20), N, (3, 64, 64), "cuda")
x = torch.randn(N, 3, 64, 64, device="cuda")
rows = bench.time_kernels(eng, x, N, it)
tot = {}
for r in rows:
    print("%-28s %-10s %-64s %7.1f us %6.1f TF%s" % (r["kernel"], r["layer"], r["instance"][:64], r["sec"] * 1e6, r["flops"] / r["sec"] / 1e12,
                                                   "  (algorithmic; winograd)" if r.get("winograd") else ""))
    k = r["kind"]
    a = tot.setdefault(k, [0.0, 0.0]); a[0] += r["sec"]; a[1] += r["flops"]
for k, (s, f) in tot.items():
    print("TOTAL %-12s %8.1f us %6.1f TF" % (k, s * 1e6, f / s / 1e12))
print("ALL %.1f us" % (sum(v[0] for v in tot.values()) * 1e6))
