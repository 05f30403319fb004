"""Dev-container harness that makes /root/reference importable on CPU.

ONLY used by tests/golden/make_golden.py (fixture generation). The reference
never travels to the GPU box; nothing under tests/ -m gpu, smoke() or bench.py
imports this.

What it does:
  * puts the torchvision / torchnet / quadprog stand-ins on sys.path
  * Tensor.cuda / Module.cuda -> identity (the reference calls .cuda() 176x)
  * TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD=1 (reference torch.load()s whole modules)
  * sys.path += /root/reference/src
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src"


def install():
    os.environ["TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD"] = "1"
    if HERE not in sys.path:
        sys.path.insert(0, HERE)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import torch
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    return torch
