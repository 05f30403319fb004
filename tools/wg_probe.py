"""Experiment: kernel-only durations (run under rocprofv3 --kernel-trace) of the weight-gradient launches of the small layers."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clsurvey_amd import ops
N = 200
for C, K, hw in ((64, 64, 16), (64, 128, 8), (128, 128, 8)):
    x = torch.randn(N, C, hw, hw, device="cuda"); dy = torch.randn(N, K, hw, hw, device="cuda")
    for _ in range(6): ops.conv3x3_wino_bwd_weight(x, dy)
    for _ in range(6): ops.conv3x3_bwd_weight(x, dy)
torch.cuda.synchronize()
