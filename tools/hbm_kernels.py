#!/usr/bin/env python
"""HBM rate of the fused optimizer / importance / gradient-memory kernels on an AlexNet-sized arena (57.8 M parameters —
the regulariser passes of BASELINE configs[1-3] are the same kernels on 0.6-9 M parameters, where they are launch-latency
bound).  Algorithmic bytes per parameter as in DESIGN.md section 4.  Prints one JSON object.
usage: hbm_kernels.py [n_params] [iters]"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clsurvey_amd import _lib, ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 57_823_240
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
t = {k: torch.rand(n, device=dev) * 1e-2 for k in ("theta", "grad", "omega", "init", "buf", "w", "out")}
G = torch.randn((6, n), device=dev)
L = _lib.lib()
ws = torch.zeros(L.clhip_gem_gram_ws(16), dtype=torch.uint8, device=dev)
gram = torch.zeros(256, dtype=torch.float64, device=dev)
rows5 = (C.c_int * 5)(0, 1, 2, 3, 4)
v5 = (C.c_float * 5)(0.5, 1.0, -0.5, 0.25, 2.0)
s = None
cases = [
    ("reg_sgd_step (EWC / MAS penalised momentum SGD)", 28, lambda: ops.reg_sgd_step(t["theta"], t["grad"], t["omega"], t["init"], t["buf"], 400.0, 1e-3, 0.9, 0.0, False)),
    ("reg_sgd_step without omega (plain momentum SGD)", 20, lambda: ops.reg_sgd_step(t["theta"], t["grad"], None, None, t["buf"], 0.0, 1e-3, 0.9, 0.0, False)),
    ("fisher_accum (EWC diag Fisher)", 12, lambda: ops.fisher_accum(t["omega"], t["grad"], 8000.0)),
    ("mas_accum (MAS omega)", 12, lambda: ops.mas_accum(t["omega"], t["grad"], 3, 200)),
    ("si_step (SI step + path integral)", 36, lambda: ops.si_step(t["theta"], t["grad"], t["omega"], t["init"], t["w"], t["buf"], 400.0, 1e-3, 0.9, 0.0, False)),
    ("si_consolidate", 28, lambda: ops.si_consolidate(t["omega"], t["w"], t["theta"], t["init"])),
    ("gem axpy (store_grad: copy)", 8, lambda: L.clhip_axpy(G[5].data_ptr(), t["grad"].data_ptr(), n, C.c_float(1.0), 1, s)),
    ("gem axpy (accumulate)", 12, lambda: L.clhip_axpy(G[5].data_ptr(), t["grad"].data_ptr(), n, C.c_float(1.0), 0, s)),
    ("gem gram, 5 rows (f64, one pass)", 20, lambda: L.clhip_gem_gram(G.data_ptr(), n, rows5, 5, n, gram.data_ptr(), ws.data_ptr(), ws.numel(), s)),
    ("gem project, 5 rows", 28, lambda: L.clhip_gem_project(G.data_ptr(), n, rows5, v5, 5, t["grad"].data_ptr(), t["out"].data_ptr(), n, s)),
]
out = {"n_params": n, "peak_TBps": 8.0, "kernels": []}
for name, bpp, fn in cases:
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    out["kernels"].append({"kernel": name, "bytes_per_param": bpp, "us": round(us, 1), "TBps": round(bpp * n / us / 1e6, 2),
                           "frac_of_8TBps": round(bpp * n / us / 1e6 / 8.0, 3)})
print(json.dumps(out))
