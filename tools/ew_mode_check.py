#!/usr/bin/env python
"""Experiment (round 4): clhip_reg_sgd_step under CLHIP_EW_MODE = 0 / 1 / 2 — time on a 57.8 M-parameter arena and a bit-level
checksum of the result (the modes must agree bit for bit: same arithmetic per element).  usage: CLHIP_EW_MODE=m ew_mode_check.py"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clsurvey_amd import ops  # noqa: E402

n = 57_823_240
g = torch.Generator(device="cuda").manual_seed(3)
t = {k: torch.rand(n, device="cuda", generator=g) * 1e-2 for k in ("theta", "grad", "omega", "init", "buf")}
for first in (True, False, False):
    ops.reg_sgd_step(t["theta"], t["grad"], t["omega"], t["init"], t["buf"], 400.0, 1e-3, 0.9, 1e-4, first)
torch.cuda.synchronize()
h = hashlib.sha256(t["theta"].cpu().numpy().tobytes() + t["buf"].cpu().numpy().tobytes()).hexdigest()[:16]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.reg_sgd_step(t["theta"], t["grad"], t["omega"], t["init"], t["buf"], 400.0, 1e-3, 0.9, 0.0, False)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
print("CLHIP_EW_MODE=%s  %.1f us  %.2f TB/s  sha %s" % (os.environ.get("CLHIP_EW_MODE", "0"), us, 28.0 * n / us / 1e6, h))
