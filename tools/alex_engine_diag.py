#!/usr/bin/env python
"""Diagnostic: plain AlexNet train-step gradients of the plan executor against torch autograd (fp64, on the device)."""
import os, sys, copy
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clsurvey_amd import models
from clsurvey_amd.net import NetEngine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
m = models.AlexNet(num_classes=20)
for mod in m.modules():
    if isinstance(mod, (torch.nn.Conv2d, torch.nn.Linear)):
        torch.nn.init.kaiming_normal_(mod.weight, nonlinearity="relu"); torch.nn.init.normal_(mod.bias, std=0.05)
ref = copy.deepcopy(m).double().cuda()
eng = NetEngine(m, N, (3, 224, 224), "cuda")
m.eval(); ref.eval()
x = torch.randn(N, 3, 224, 224, device="cuda"); y = torch.randint(0, 20, (N,), device="cuda")
eng.loss_step(x, y, "ce_mean", True)
out = ref.classifier(torch.flatten(ref.features(x.double()), 1))
F.cross_entropy(out, y).backward()
for (n, p), q in zip(m.named_parameters(), ref.parameters()):
    a, b = p.grad.double(), q.grad
    print("%-22s max %.2e l2 %.2e" % (n, float((a - b).abs().max() / b.abs().max()), float((a - b).norm() / b.norm())))
