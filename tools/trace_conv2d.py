#!/usr/bin/env python
"""Per-wave cycle sums of the gather-GEMM loop phases (tuning aid; needs the -DCLHIP_TRACE variant of conv2d.hip:
python tools/mkvars.py conv2d.hip c2trace=CLHIP_TRACE; CLHIP_LIB=clsurvey_amd/libclhip_c2trace.so python tools/trace_conv2d.py)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clsurvey_amd import _lib, ops  # noqa: E402

L = _lib.lib()
L.clhip_debug_set_conv2d_trace.restype = C.c_int
L.clhip_debug_set_conv2d_trace.argtypes = [C.c_void_p]
buf = torch.zeros(4096 * 4 * 8, dtype=torch.int64, device="cuda")
N = 128
x = torch.randn(N, 64, 27, 27, device="cuda"); w = torch.randn(192, 64, 5, 5, device="cuda") * 0.05; b = torch.zeros(192, device="cuda")
y = ops.conv2d_fwd(x, w, b, 1, 2, True); dy = torch.randn_like(y)
x1 = torch.randn(N, 3, 224, 224, device="cuda"); w1 = torch.randn(64, 3, 11, 11, device="cuda") * 0.05; b1 = torch.zeros(64, device="cuda")
cases = {"conv2 fwd": lambda: ops.conv2d_fwd(x, w, b, 1, 2, True), "conv2 dgrad": lambda: ops.conv2d_bwd_data(dy, w, x.shape, 1, 2, x),
         "conv2 wgrad": lambda: ops.conv2d_bwd_weight(x, dy, (5, 5), 1, 2), "conv1 fwd": lambda: ops.conv2d_fwd(x1, w1, b1, 4, 2, True)}
for name, fn in cases.items():
    fn(); fn(); torch.cuda.synchronize()
    buf.zero_(); assert L.clhip_debug_set_conv2d_trace(buf.data_ptr()) == 0
    fn(); torch.cuda.synchronize(); L.clhip_debug_set_conv2d_trace(None)
    t = buf.cpu().numpy().reshape(4096, 4, 8).astype(np.int64)
    t = t[t[:, 0, 0] != 0]
    tot = (t[:, :, 1] - t[:, :, 0]).astype(np.float64)
    ph = {k: t[:, :, i].astype(np.float64) for k, i in (("barrier1", 2), ("lds_store(+wait loads)", 3), ("barrier2", 4), ("load issue", 5), ("mfma loop", 6))}
    print("%-12s blocks traced %4d  chunks %3d  loop cycles/wave %9.0f | " % (name, t.shape[0], int(t[0, 0, 7]), tot.mean()) +
          "  ".join("%s %4.1f%%" % (k, 100 * v.sum() / tot.sum()) for k, v in ph.items()))
