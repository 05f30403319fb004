"""Time the bf16-split conv entry points on ONE layer shape (HIP events, best of 3 x 20 launches).
usage: python tools/bs_layer.py N C K HW"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clsurvey_amd import ops  # noqa: E402
from tools.bs_bench import timed  # noqa: E402

N, C, K, HW = (int(v) for v in sys.argv[1:5])
dev = torch.device("cuda:0")
x = torch.randn(N, C, HW, HW, device=dev).relu_()
w = torch.randn(K, C, 3, 3, device=dev) * (2.0 / (9 * C)) ** 0.5
b = torch.randn(K, device=dev) * 0.1
dy = torch.randn(N, K, HW, HW, device=dev)
yp, idx = ops.conv3x3_relu_pool_fwd(x, w, b)
dyp = torch.randn_like(yp)
t = [timed(lambda: ops.conv3x3_bs_fwd(x, w, b, True, pool=True)), timed(lambda: ops.conv3x3_bs_fwd(x, w, b, True)),
     timed(lambda: ops.conv3x3_bs_bwd_data(dy, w, x)), timed(lambda: ops.conv3x3_bs_bwd_data(dyp, w, None, idx))]
print("%-28s %dx%d@%d N=%d  fwd+pool %6.1f  fwd %6.1f  bwd-data(mask) %6.1f  bwd-data(unpool) %6.1f us"
      % (os.path.basename(os.environ.get("CLHIP_LIB", "libclhip.so")), C, K, HW, N, t[0], t[1], t[2], t[3]))
