#!/usr/bin/env python
"""HIP-event timing of the gather-GEMM convolutions at AlexNet's layer shapes (N = 128). usage: conv2d_bench.py [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clsurvey_amd import ops
it = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = 128
LAYERS = [("conv1 11x11/4 3->64 @224", 3, 64, 224, 11, 4, 2), ("conv2 5x5 64->192 @27", 64, 192, 27, 5, 1, 2),
          ("conv3 3x3 192->384 @13", 192, 384, 13, 3, 1, 1), ("conv4 3x3 384->256 @13", 384, 256, 13, 3, 1, 1),
          ("conv5 3x3 256->256 @13", 256, 256, 13, 3, 1, 1)]


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for name, C, K, H, R, st, pad in LAYERS:
    x = torch.randn(N, C, H, H, device="cuda"); w = torch.randn(K, C, R, R, device="cuda") * 0.05; b = torch.zeros(K, device="cuda")
    y = ops.conv2d_fwd(x, w, b, st, pad, True)
    dy = torch.randn_like(y)
    fl = 2.0 * R * R * C * K * y.shape[2] * y.shape[3] * N
    row = "%-26s" % name
    if R != 3:
        t = timed(lambda: ops.conv2d_fwd(x, w, b, st, pad, True)); row += " fwd %7.1f us %5.1f TF" % (t, fl / t / 1e6)
        if C > 3:
            t = timed(lambda: ops.conv2d_bwd_data(dy, w, x.shape, st, pad, x)); row += "  dgrad %7.1f us %5.1f TF" % (t, fl / t / 1e6)
    t = timed(lambda: ops.conv2d_bwd_weight(x, dy, (R, R), st, pad)); row += "  wgrad %7.1f us %5.1f TF" % (t, fl / t / 1e6)
    print(row)
