#!/usr/bin/env python
"""Per-parameter gradient error table (GPU vs fp64 oracle vs fp32 CPU oracle).
usage: grad_table.py [N] [small_VGG9|base_VGG9|wide_VGG9|deep_VGG22] [param indices for the per-row listing]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vgg_ref  # noqa: E402
from clsurvey_amd import models, net  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
NAME = sys.argv[2] if len(sys.argv) > 2 else "small_VGG9"
cfg = vgg_ref.CFGS[NAME]
FC = (128, 128) if NAME == "small_VGG9" else (512, 512)
params = vgg_ref.init_params(cfg, FC, 20, 64, np.random.RandomState(21))
gen = np.random.RandomState(22)
x = torch.from_numpy(gen.standard_normal((N, 3, 64, 64)).astype(np.float32))
y = torch.from_numpy(gen.randint(0, 20, size=(N,)).astype(np.int64))
torch.set_num_threads(32)
_, _, g32, _ = vgg_ref.loss_and_grads(params, cfg, x, y, "ce_sum")
_, _, g64, _ = vgg_ref.loss_and_grads([p.double() for p in params], cfg, x.double(), y, "ce_sum")
npool = sum(1 for v in cfg if v == "M")
m = models.VGGSlim(cfg=cfg, num_classes=20, classifier_inputdim=[v for v in cfg if v != "M"][-1] * (64 // 2 ** npool) ** 2,
                   classifier_dim1=FC[0], classifier_dim2=FC[1])
with torch.no_grad():
    for p, q in zip(m.parameters(), params):
        p.copy_(q)
eng = net.NetEngine(m, N, (3, 64, 64), "cuda:0")
eng.loss_step(x.cuda(), y.cuda(), "ce_sum", True)
torch.cuda.synchronize()


def rel(a, b):
    return float((a.double().cpu() - b.double()).abs().max() / b.double().abs().max())


print("%3s %-22s %10s %10s %10s" % ("i", "shape", "gpu-f64", "cpu32-f64", "gpu-cpu32"))
for i, (p, a, b) in enumerate(zip(m.parameters(), g32, g64)):
    print("%3d %-22s %10.2e %10.2e %10.2e" % (i, tuple(p.shape), rel(p.grad, b), rel(a, b), rel(p.grad, a)))
b1 = g64[1]
print("conv1 bias: gpu", p_ := list(m.parameters())[1].grad[:6].cpu().tolist())
print("conv1 bias: f64", b1[:6].tolist())
print("conv1 bias: c32", g32[1][:6].tolist())
for i in ((0, 2, 4) if len(sys.argv) < 4 else [int(v) for v in sys.argv[3].split(",")]):
    g = list(m.parameters())[i].grad.double().cpu()
    d = (g - g64[i]).abs().reshape(g.shape[0], -1).max(1).values / g64[i].abs().max()
    top = torch.topk(d, 6)
    print("param %d: rows with largest error:" % i, [(int(k), "%.1e" % v) for v, k in zip(top.values, top.indices)],
          " median row err %.1e" % d.median())
