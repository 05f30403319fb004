#!/usr/bin/env python
"""Gradient arenas of a few plan-executor steps, to compare two builds of the library bit for bit (CLHIP_LIB selects the build):
    python tools/experiments/pair_engine_check.py dump out.pt      |      ... cmp a.pt b.pt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    for k in a:
        print("%-44s %s  |grad| %.6g" % (k, "identical" if k not in bad else "DIFFERENT max %.3g" % (a[k] - b[k]).abs().max().item(),
                                          a[k].norm().item()))
    sys.exit(1 if bad else 0)

from clsurvey_amd import models  # noqa: E402
from clsurvey_amd.net import NetEngine  # noqa: E402

out = {}
for name, hw, N in (("small_VGG9_cl_128_128", 64, 200), ("small_VGG9_cl_128_128", 64, 37), ("base_VGG9_cl_512_512", 64, 200),
                    ("wide_VGG9_cl_512_512", 64, 50), ("small_VGG9_cl_128_128_DROP_BN", 64, 40), ("deep_VGG22_cl_512_512", 64, 24)):
    torch.manual_seed(3)
    m = models.parse_model_name(name, (hw, hw), 20)
    eng = NetEngine(m, N, (3, hw, hw), "cuda")
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    x = torch.randn((N, 3, hw, hw), generator=g, device="cuda")
    y = torch.randint(0, 20, (N,), generator=g, device="cuda")
    eng.loss_step(x, y, "ce_mean", True)
    out["%s N=%d" % (name, N)] = eng.arena.grad.detach().clone().cpu()
    del eng
torch.save(out, sys.argv[2])
print("saved", len(out), "gradient arenas to", sys.argv[2], "from", os.path.basename(os.environ.get("CLHIP_LIB", "libclhip.so")))
