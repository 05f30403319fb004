"""Experiment: what does the ReLU-mask read in the backward-data epilogue cost?  Times backward-data (Winograd where the plan
uses it, direct otherwise) with and without relu_src on the bench model's layers whose input is a pooled tensor."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clsurvey_amd import ops

def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

d = "cuda"
N = 200
for (C, K, H, unpool, wino) in [(64, 64, 32, True, True), (64, 64, 16, False, True), (64, 128, 8, False, False), (128, 128, 8, True, False), (64, 64, 16, True, True)]:
    w = torch.randn(K, C, 3, 3, device=d) * 0.05
    x = torch.randn(N, C, H, H, device=d).relu()
    if unpool:
        dy = torch.randn(N, K, H // 2, H // 2, device=d)
        idx = torch.randint(0, 4, (N, K, H // 2, H // 2), device=d, dtype=torch.uint8)
    else:
        dy, idx = torch.randn(N, K, H, H, device=d), None
    if wino:
        f = lambda m: ops.conv3x3_wino_bwd_data(dy, w, m, idx)
    elif unpool:
        f = lambda m: ops.conv3x3_bwd_data_unpool(dy, idx, w, m)
    else:
        f = lambda m: ops.conv3x3_bwd_data(dy, w, m)
    print("%3dx%-3d @%-2d unpool=%d wino=%d   with mask %7.1f us   without %7.1f us" % (C, K, H, unpool, wino, t(lambda: f(x)), t(lambda: f(None))), flush=True)
