#!/usr/bin/env python
"""Run ONE conv entry point a few times (for rocprofv3 --pmc passes).
usage: python tools/one_kernel.py {c3pool|fwdpool|fwd|dgrad|dgrad_unpool|wgrad|wino_fwdpool|wino_fwd|wino_dgrad|wino_dgrad_unpool|wino_dgrad_unpool_mask|
wino_wgrad|wino_wgrad_unpool|bs_wgrad|bs_wgrad_unpool|bs_fwdpool|bs_fwd|bs_dgrad|bs_dgrad_unpool} N C K HW [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clsurvey_amd import ops  # noqa: E402

kind = sys.argv[1]
N, C, K, HW = (int(v) for v in sys.argv[2:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device("cuda:0")
x = torch.randn(N, C, HW, HW, device=dev)
w = torch.randn(K, C, 3, 3, device=dev) * 0.05
b = torch.zeros(K, device=dev)
dy = torch.randn(N, K, HW, HW, device=dev)
yp, idx = ops.conv3x3_relu_pool_fwd(x, w, b)
dyp = torch.randn_like(yp)
fn = {"dgrad_unpool": lambda: ops.conv3x3_bwd_data_unpool(dyp, idx, w, x), "c3pool": lambda: ops.conv3x3_relu_pool_fwd(x, w, b), "fwdpool": lambda: ops.conv3x3_relu_pool_fwd(x, w, b), "fwd": lambda: ops.conv3x3_fwd(x, w, b, True),
      "dgrad": lambda: ops.conv3x3_bwd_data(dy, w, x), "wgrad": lambda: ops.conv3x3_bwd_weight(x, dy),
      "wino_fwdpool": lambda: ops.conv3x3_wino_fwd(x, w, b, True, pool=True), "wino_fwd": lambda: ops.conv3x3_wino_fwd(x, w, b, True),
      "wino_dgrad": lambda: ops.conv3x3_wino_bwd_data(dy, w, x), "wino_dgrad_unpool": lambda: ops.conv3x3_wino_bwd_data(dyp, w, None, idx),      # as the plan calls it: no mask read behind a fused pool (common.hpp, CLHIP_POOL_DEAD)
      "wino_dgrad_unpool_mask": lambda: ops.conv3x3_wino_bwd_data(dyp, w, x, idx),
      "bs_fwdpool": lambda: ops.conv3x3_bs_fwd(x, w, b, True, pool=True), "bs_fwd": lambda: ops.conv3x3_bs_fwd(x, w, b, True),
      "bs_dgrad": lambda: ops.conv3x3_bs_bwd_data(dy, w, x), "bs_dgrad_unpool": lambda: ops.conv3x3_bs_bwd_data(dyp, w, None, idx),
      "bs_wgrad": lambda: ops.conv3x3_bs_bwd_weight(x, dy), "bs_wgrad_unpool": lambda: ops.conv3x3_bs_bwd_weight(x, dyp, idx),
      "wino_wgrad": lambda: ops.conv3x3_wino_bwd_weight(x, dy), "wino_wgrad_unpool": lambda: ops.conv3x3_wino_bwd_weight(x, dyp, idx)}[kind]
for _ in range(iters):
    fn()
torch.cuda.synchronize()
