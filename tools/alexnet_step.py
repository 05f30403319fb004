#!/usr/bin/env python
"""AlexNet (224x224) train steps on the plan executor: ms / step and images / s; run under rocprofv3 --kernel-trace
--stats for the per-kernel split.  usage: alexnet_step.py [batch] [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clsurvey_amd import models
from clsurvey_amd.net import NetEngine
from clsurvey_amd.optim import SGD
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
m = models.parse_model_name("alexnet_scratch", num_classes=200)
eng = NetEngine(m, N, (3, 224, 224), "cuda")
opt = SGD(m.parameters(), 0.01, momentum=0.9)
m.train()
x = torch.randn(N, 3, 224, 224, device="cuda"); y = torch.randint(0, 200, (N,), device="cuda")
for _ in range(2):
    eng.loss_step(x, y, "ce_mean", True); opt.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    eng.loss_step(x, y, "ce_mean", True); opt.step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
# 2*MAC: forward 1428.3 MFLOP / image; train = 3 * fwd - conv1 backward-data
print("alexnet N=%d: %.3f ms/step, %.0f img/s, %.1f TFLOP/s" % (N, dt * 1e3, N / dt, N * (3 * 1428.3e6 - 140.6e6) / dt / 1e12))
