#!/usr/bin/env python
"""Loss, logits and the whole gradient arena of a few EWC-style passes of the bench model on fixed inputs, saved to argv[1] (two builds
compared bitwise across processes; CLHIP_LIB selects the build)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clsurvey_amd import models, net, ops
out = {}
for name, N, hw in (("small_VGG9_cl_128_128", 200, 64), ("small_VGG9_cl_128_128", 37, 64), ("base_VGG9_cl_512_512", 50, 64)):
    torch.manual_seed(11)
    m = models.parse_model_name(name, (hw, hw), 20)
    eng = net.NetEngine(m, N, (3, hw, hw), "cuda")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, 3, hw, hw, generator=g).cuda(); y = torch.randint(0, 20, (N,), generator=g).cuda()
    for kind in ("ce_mean", "ce_sum"):
        eng.arena.grad.zero_()
        loss, logits = eng.loss_step(x, y, kind, backward=True, want_logits=True)
        torch.cuda.synchronize()
        tag = "%s_%d_%s" % (name, N, kind)
        out[tag + "_loss"] = loss.cpu().numpy(); out[tag + "_logits"] = logits.cpu().numpy(); out[tag + "_grad"] = eng.arena.grad.cpu().numpy()
np.savez(sys.argv[1], **out)
print("saved", len(out))
