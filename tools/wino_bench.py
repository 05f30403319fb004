#!/usr/bin/env python
"""Winograd F(2x2, 3x3) kernels against the direct MFMA kernels, per small_VGG9 / wide_VGG9 layer shape at N = 200
(HIP events, weight transform included in the Winograd time).  usage: wino_bench.py [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clsurvey_amd import ops
it = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ALEX = len(sys.argv) > 2 and sys.argv[2] == "alex"        # AlexNet's 13x13 layers at N = 128 (weight gradient: the gather-GEMM is the direct path)
N = 128 if ALEX else 200


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it


SHAPES = ((192, 384, 13), (384, 256, 13), (256, 256, 13)) if ALEX else \
    ((64, 64, 32), (64, 64, 16), (64, 128, 8), (128, 128, 8), (64, 128, 32), (128, 256, 16), (256, 256, 16), (256, 512, 8), (512, 512, 8))
for C, K, hw in SHAPES:
    x = torch.randn(N, C, hw, hw, device="cuda"); w = torch.randn(K, C, 3, 3, device="cuda") * 0.05; b = torch.randn(K, device="cuda")
    dy = torch.randn(N, K, hw, hw, device="cuda")
    fl = 2.0 * 9 * C * K * hw * hw * N
    rows = [("fwd", lambda: ops.conv3x3_fwd(x, w, b, True), lambda: ops.conv3x3_wino_fwd(x, w, b, True))]
    if not hw & 1:
        rows.append(("fwd+pool", lambda: ops.conv3x3_relu_pool_fwd(x, w, b), lambda: ops.conv3x3_wino_fwd(x, w, b, True, pool=True)))
    rows.append(("bwd_data", lambda: ops.conv3x3_bwd_data(dy, w, x), lambda: ops.conv3x3_wino_bwd_data(dy, w, x)))
    if C % 64 == 0 and K % 64 == 0:
        rows.append(("bwd_weight", (lambda: ops.conv2d_bwd_weight(x, dy, (3, 3), 1, 1)) if ALEX else (lambda: ops.conv3x3_bwd_weight(x, dy)),
                     lambda: ops.conv3x3_wino_bwd_weight(x, dy)))
    out = "%4dx%-4d@%-3d" % (C, K, hw)
    for name, direct, wino in rows:
        td, tw = timed(direct), timed(wino)
        out += "  %s direct %6.1f us (%5.1f TF)  wino %6.1f us (%5.1f TF-equivalent) x%.2f" % (name, td, fl / td / 1e6, tw, fl / tw / 1e6, td / tw)
    print(out, flush=True)
