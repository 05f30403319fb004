"""5x5 / padding-2 weight gradient: bf16-split kernel (csrc/bswgrad5.hip, slabs + reduction) against the gather-GEMM of conv2d.hip, HIP events,
best of 3 x 20 launches.  usage: python tools/bs_wgrad5_bench.py N C K HW [...]"""
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
    fl = 2.0 * 25 * C * K * HW * HW * N
    t = [timed(lambda: ops.conv5x5_bs_bwd_weight(x, dy)), timed(lambda: ops.conv2d_bwd_weight(x, dy, (5, 5), 1, 2))]
    print("5x5 %dx%d@%d N=%d  bf16-split %6.1f us = %5.1f TF   gather-GEMM %6.1f us = %5.1f TF" % (C, K, HW, N, t[0], fl / t[0] / 1e6, t[1], fl / t[1] / 1e6), flush=True)
