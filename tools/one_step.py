#!/usr/bin/env python
"""A few plan-executor loss_steps of a VGG config at batch 200 (for rocprofv3 passes). usage: one_step.py [steps] [model]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clsurvey_amd import models
from clsurvey_amd.net import NetEngine
name = sys.argv[2] if len(sys.argv) > 2 else "small_VGG9_cl_128_128"
m = models.parse_model_name(name, (64, 64), 20)
eng = NetEngine(m, 200, (3, 64, 64), "cuda")
x = torch.randn(200, 3, 64, 64, device="cuda"); y = torch.randint(0, 20, (200,), device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    eng.loss_step(x, y, "ce_mean", True)
torch.cuda.synchronize()
