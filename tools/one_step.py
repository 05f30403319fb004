#!/usr/bin/env python
"""A few engine loss_steps on the bench model (for rocprofv3 --pmc passes over the non-conv kernels)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clsurvey_amd import models
from clsurvey_amd.net import NetEngine
m = models.parse_model_name("small_VGG9_cl_128_128", (64, 64), 20)
eng = NetEngine(m, 200, (3, 64, 64), "cuda")
x = torch.randn(200, 3, 64, 64, device="cuda"); y = torch.randint(0, 20, (200,), device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    eng.loss_step(x, y, "ce_mean", True)
torch.cuda.synchronize()
