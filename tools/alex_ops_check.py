#!/usr/bin/env python
"""Every AlexNet-shaped operator of the library against torch CPU at a given batch (default 4): relative max error per output."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clsurvey_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
g = torch.Generator().manual_seed(3)


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


for name, C, K, H, R, st, pad in [("conv1", 3, 64, 224, 11, 4, 2), ("conv2", 64, 192, 27, 5, 1, 2), ("conv3", 192, 384, 13, 3, 1, 1),
                                  ("conv4", 384, 256, 13, 3, 1, 1), ("conv5", 256, 256, 13, 3, 1, 1)]:
    x = torch.randn(N, C, H, H, generator=g); w = torch.randn(K, C, R, R, generator=g) * 0.05; b = torch.randn(K, generator=g) * 0.1
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    y = F.relu(F.conv2d(xr, wr, br, stride=st, padding=pad))
    dy = torch.randn(y.shape, generator=g) * (y > 0)
    y.backward(dy)
    xd, wd, bd, dyd = x.cuda(), w.cuda(), b.cuda(), dy.cuda()
    if R == 3:
        yd = ops.conv3x3_fwd(xd, wd, bd, True)
        dxd = ops.conv3x3_bwd_data(dyd, wd)
        dwd, dbd = ops.conv2d_bwd_weight(xd, dyd, (3, 3), 1, 1)
        dwd2, dbd2 = ops.conv3x3_bwd_weight(xd, dyd)
        print("%s conv3x3 wgrad (3x3 kernel) dw %.1e db %.1e" % (name, rel(dwd2, wr.grad), rel(dbd2, br.grad)))
    else:
        yd = ops.conv2d_fwd(xd, wd, bd, st, pad, True)
        dxd = ops.conv2d_bwd_data(dyd, wd, x.shape, st, pad)
        dwd, dbd = ops.conv2d_bwd_weight(xd, dyd, (R, R), st, pad)
    print("%s fwd %.1e  dgrad %.1e  wgrad %.1e  db %.1e" % (name, rel(yd, y), rel(dxd, xr.grad), rel(dwd, wr.grad), rel(dbd, br.grad)))
for C, H in ((64, 55), (192, 27), (256, 13)):
    x = torch.randn(N, C, H, H, generator=g).requires_grad_(True)
    y = F.max_pool2d(x, 3, 2)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    yd, idx = ops.maxpool_fwd(x.detach().cuda(), 3, 2)
    dxd = ops.maxpool_bwd(dy.cuda(), idx, x.shape, 3, 2)
    print("pool %dx%d fwd %.1e bwd %.1e" % (H, H, rel(yd, y), rel(dxd, x.grad)))
