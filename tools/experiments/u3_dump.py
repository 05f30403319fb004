#!/usr/bin/env python
"""Weight / bias gradient of the first layer through clhip_conv3x3_bwd_weight_unpool on fixed inputs, saved to argv[1] (round 6, second
session: two builds of conv3x3_wgrad_c3_unpool_kernel compared bitwise across processes: CLHIP_LIB selects the build)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clsurvey_amd import ops
out = {}
for (N, C, K, H, W) in [(200, 3, 64, 64, 64), (7, 3, 64, 32, 64), (5, 3, 40, 6, 32), (3, 2, 64, 8, 96), (2, 1, 33, 4, 32)]:
    g = torch.Generator().manual_seed(N * 1000 + K)
    x = torch.randn(N, C, H, W, generator=g).cuda()
    gp = torch.randn(N, K, H // 2, W // 2, generator=g).cuda()
    idx = torch.randint(0, 5, (N, K, H // 2, W // 2), generator=g, dtype=torch.uint8).cuda()
    dw, db = ops.conv3x3_bwd_weight_unpool(x, gp, idx)
    dy = ops.maxpool2_bwd(gp, idx)
    dw_ref, db_ref = ops.conv3x3_bwd_weight(x, dy)
    db64 = dy.double().sum((0, 2, 3))
    tag = "%dx%dx%dx%dx%d" % (N, C, K, H, W)
    out[tag + "_dw"] = dw.cpu().numpy(); out[tag + "_db"] = db.cpu().numpy()
    print(tag, "dw vs unfused max|d| %.3e" % float((dw - dw_ref).abs().max()), "db err vs f64: fused %.3e unfused %.3e (|db| max %.3e)" % (
        float((db.double() - db64).abs().max()), float((db_ref.double() - db64).abs().max()), float(db64.abs().max())))
np.savez(sys.argv[1], **out)
