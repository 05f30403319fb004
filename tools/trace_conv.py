#!/usr/bin/env python
"""Cycle-stamp trace of the chunked conv kernels (tuning aid, needs a -DCLHIP_TRACE variant library:
python -c "from clsurvey_amd import build as b; b.build_variant('trace', ['conv3x3.hip','conv3x3_wgrad.hip'], ['CLHIP_TRACE'])"
CLHIP_LIB=clsurvey_amd/libclhip_trace.so python tools/trace_conv.py out.npz)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clsurvey_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
for nm in ("clhip_debug_set_conv_trace", "clhip_debug_set_wgrad_trace"):
    f = getattr(L, nm)
    f.restype = C.c_int
    f.argtypes = [C.c_void_p]
MAXB = 1 << 16
buf = torch.zeros(MAXB * 4 * 16, dtype=torch.int64, device=dev)
out = {}


def run(tag, setter, fn):
    fn(); fn()
    torch.cuda.synchronize()
    buf.zero_()
    assert setter(buf.data_ptr()) == 0
    fn()
    torch.cuda.synchronize()
    setter(None)
    t = buf.cpu().numpy().reshape(MAXB, 4, 16)
    nb = int((t[:, 0, 0] != 0).sum())
    t = t[:nb].astype(np.int64)
    out[tag] = t
    t0 = t[:, :, 0].min()
    st, idx, pro, loop, end = [t[:, :, i] - t0 for i in range(5)]
    ld, mf, stt, ba = [t[:, :, i] for i in (5, 6, 7, 8)]
    cu = (t[:, 0, 10] & 0xf) * 4096 + ((t[:, 0, 9] >> 8) & 0xfff)      # (xcc, se/sh/cu bits of HW_ID)
    ncu = len(np.unique(cu))
    cnt = np.unique(cu, return_counts=True)[1]
    print("%-28s blocks %5d  CUs %3d (max %d/CU)  makespan %6d cyc | start p50 %6d max %6d | "
          "idx %5d pro %5d loop %6d [ld %5d mfma %6d store %5d barrier %5d] epi %5d | nchunk %d" % (
              tag, nb, ncu, cnt.max(), end.max(), np.median(st), st.max(),
              np.median(idx - st), np.median(pro - idx), np.median(loop - pro),
              np.median(ld), np.median(mf), np.median(stt), np.median(ba), np.median(end - loop), t[0, 0, 11]))


def conv_case(c, k, hw, N=200):
    x = torch.randn(N, c, hw, hw, device=dev)
    w = torch.randn(k, c, 3, 3, device=dev) * 0.05
    b = torch.zeros(k, device=dev)
    dy = torch.randn(N, k, hw, hw, device=dev)
    tag = "%dx%d@%d" % (c, k, hw)
    run(tag + " fwd", L.clhip_debug_set_conv_trace, lambda: ops.conv3x3_fwd(x, w, b, True))
    run(tag + " bwd_data", L.clhip_debug_set_conv_trace, lambda: ops.conv3x3_bwd_data(dy, w, x))
    run(tag + " wgrad", L.clhip_debug_set_wgrad_trace, lambda: ops.conv3x3_bwd_weight(x, dy))


cases = [(64, 64, 32), (64, 64, 16), (64, 128, 8), (128, 128, 8), (128, 256, 8), (256, 256, 16)]
for c in cases:
    conv_case(*c)
if len(sys.argv) > 1:
    np.savez_compressed(sys.argv[1], **{k.replace(" ", "_"): v for k, v in out.items()})
