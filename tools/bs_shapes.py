"""bf16-split vs Winograd conv kernels on a list of (N, C, K, H, W) shapes: forward (+ReLU) and backward-data (+mask), HIP events.
usage: python tools/bs_shapes.py alex|wide224|<N,C,K,H,W> ..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clsurvey_amd import ops  # noqa: E402
from tools.bs_bench import timed  # noqa: E402

SETS = {"alex": [(128, 192, 384, 13, 13), (128, 384, 256, 13, 13), (128, 256, 256, 13, 13)],
        "wide224": [(50, 64, 128, 112, 112), (50, 128, 256, 56, 56), (50, 256, 256, 56, 56), (50, 256, 512, 28, 28), (50, 512, 512, 28, 28)]}
shapes = []
for a in sys.argv[1:] or ["alex"]:
    shapes += SETS[a] if a in SETS else [tuple(int(v) for v in a.split(","))]
dev = torch.device("cuda:0")
for N, C, K, H, W in shapes:
    x = torch.randn(N, C, H, W, device=dev).relu_()
    w = torch.randn(K, C, 3, 3, device=dev) * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(K, device=dev) * 0.1
    dy = torch.randn(N, K, H, W, device=dev)
    fl = 2.0 * 9 * C * K * H * W * N
    row = []
    for name, f, d in (("wino", ops.conv3x3_wino_fwd, ops.conv3x3_wino_bwd_data), ("bs", ops.conv3x3_bs_fwd, ops.conv3x3_bs_bwd_data)):
        try:
            tf = timed(lambda: f(x, w, b, True))
            td = timed(lambda: d(dy, w, x)) if (C % 64 == 0 or name == "wino") else float("nan")
        except Exception as e:          # shape outside the path's domain
            tf = td = float("nan")
        row.append((name, tf, td))
    print("%4dx%3d->%3d @%3dx%3d  %s   floor(bs) %.1f us" % (N, C, K, H, W, "   ".join("%s fwd %7.1f bwd %7.1f" % r for r in row), fl * 6 / 2.5e15 * 1e6))
