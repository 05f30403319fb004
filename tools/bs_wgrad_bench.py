"""Weight gradient of one 3x3 layer: bf16-split kernel (csrc/bswgrad.hip) against the Winograd f32 kernel, HIP events, best of 3 x 20 launches,
each including its slab reduction.  usage: python tools/bs_wgrad_bench.py N C K HW [N C K HW ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clsurvey_amd import ops  # noqa: E402
from tools.bs_bench import timed  # noqa: E402

a = [int(v) for v in sys.argv[1:]]
for i in range(0, len(a), 4):
    N, C, K, HW = a[i:i + 4]
    dev = torch.device("cuda:0")
    x = torch.randn(N, C, HW, HW, device=dev).relu_()
    dy = torch.randn(N, K, HW, HW, device=dev)
    dyp = torch.randn(N, K, HW // 2, HW // 2, device=dev)
    idx = torch.randint(0, 5, (N, K, HW // 2, HW // 2), device=dev, dtype=torch.uint8)
    fl = 2.0 * 9 * C * K * HW * HW * N
    pooled = HW % 16 == 0 and C % 64 == 0 and K % 64 == 0
    t = [timed(lambda: ops.conv3x3_bs_bwd_weight(x, dy)), timed(lambda: ops.conv3x3_bs_bwd_weight(x, dyp, idx)) if pooled else float("nan"),
         timed(lambda: ops.conv3x3_wino_bwd_weight(x, dy)), timed(lambda: ops.conv3x3_wino_bwd_weight(x, dyp, idx)) if pooled else float("nan")]
    print("%dx%d@%d N=%d  bf16-split %6.1f (pooled dy %6.1f) us = %5.1f TF   Winograd %6.1f (pooled dy %6.1f) us = %5.1f TF"
          % (C, K, HW, N, t[0], t[1], fl / t[0] / 1e6, t[2], t[3], fl / t[2] / 1e6), flush=True)
