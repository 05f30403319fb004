"""Strided convolution by space-to-depth on the dense 3x3 kernels (csrc/s2dconv.hip) against torch CPU: nn.Conv2d(3, 64, 11, 4, 2) of
torchvision's AlexNet (the reference's models/net.py:96-125) forward (+ bias, ReLU) and autograd's weight / bias gradients; a stride-2
instance; the plan executor taking the path for AlexNet's first layer (CLHIP_S2D=1; off by default: measured slower) and
agreeing with the gather-GEMM path."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHAPES = [  # N, C, H, W, K, R, stride, pad
    (2, 3, 224, 224, 64, 11, 4, 2), (128, 3, 224, 224, 64, 11, 4, 2), (5, 3, 64, 64, 64, 11, 4, 2), (3, 3, 99, 83, 64, 12, 4, 0),
    (4, 8, 40, 40, 64, 5, 2, 2), (3, 16, 33, 47, 128, 6, 2, 1), (7, 1, 50, 50, 64, 9, 4, 3),
]


def _rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / max(float(b.double().abs().max()), 1e-30))


@pytest.mark.parametrize("shape", SHAPES)
def test_s2d_forward_and_weight_gradient(shape):
    from clsurvey_amd import ops
    N, C, H, W, K, R, st, pad = shape
    gen = np.random.RandomState(sum(shape))
    x = torch.from_numpy(gen.standard_normal((N, C, H, W)).astype(np.float32))
    w = torch.from_numpy((gen.standard_normal((K, C, R, R)) / np.sqrt(C * R * R)).astype(np.float32))
    b = torch.from_numpy((gen.standard_normal((K,)) * 0.1).astype(np.float32))
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    z_ref = F.conv2d(x, wr, br, stride=st, padding=pad)
    dy = torch.from_numpy(gen.standard_normal(tuple(z_ref.shape)).astype(np.float32))
    z_ref.backward(dy)
    xd, wd, bd, dyd = x.cuda(), w.cuda(), b.cuda(), dy.cuda()
    got = ops.conv2d_s2d_fwd(xd, wd, bd, st, pad, relu=False)
    assert got is not None, "the space-to-depth path should take this shape"
    z, ws = got
    assert z.shape == z_ref.shape
    assert _rel(z, z_ref.detach()) <= 2e-5
    y, _ = ops.conv2d_s2d_fwd(xd, wd, bd, st, pad, relu=True, ws=ws)
    assert _rel(y, F.relu(z_ref.detach())) <= 2e-5
    # weight gradient from the phase planes the forward call left in ws, and from x again; both deterministic
    dw, db = ops.conv2d_s2d_bwd_weight(None, dyd, (N, C, H, W), R, st, pad, ws=ws)
    assert _rel(dw, wr.grad) <= 5e-5 and _rel(db, br.grad) <= 5e-5, (_rel(dw, wr.grad), _rel(db, br.grad))
    dw2, db2 = ops.conv2d_s2d_bwd_weight(xd, dyd, (N, C, H, W), R, st, pad)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    # ... and the gather-GEMM entry points of the same layer agree with it (two summation orders of the same products)
    assert _rel(ops.conv2d_fwd(xd, wd, bd, st, pad, relu=False), z) <= 2e-5
    dwg, dbg = ops.conv2d_bwd_weight(xd, dyd, (R, R), st, pad)
    assert _rel(dwg, dw) <= 5e-5 and _rel(dbg, db) <= 5e-5


def test_s2d_declines_other_shapes():
    from clsurvey_amd import _lib
    L = _lib.lib()
    assert L.clhip_conv2d_s2d_ws(8, 3, 224, 224, 64, 11, 4, 2) > 0
    assert L.clhip_conv2d_s2d_ws(8, 3, 224, 224, 64, 7, 4, 2) == 0        # 7 <= 2 * 4: two taps, not three
    assert L.clhip_conv2d_s2d_ws(8, 3, 224, 224, 64, 13, 4, 2) == 0       # 13 > 3 * 4
    assert L.clhip_conv2d_s2d_ws(8, 8, 224, 224, 64, 11, 4, 2) == 0       # 8 * 16 phase planes > 64
    assert L.clhip_conv2d_s2d_ws(8, 3, 224, 224, 96, 11, 4, 2) == 0       # output channels not whole 64-groups
    assert L.clhip_conv2d_s2d_ws(8, 64, 27, 27, 192, 5, 1, 2) == 0        # stride 1
    x = torch.zeros(2, 8, 32, 32, device="cuda")
    assert L.clhip_conv2d_s2d_fwd(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), 2, 8, 32, 32, 64, 11, 4, 2, 0, x.data_ptr(), 1 << 20,
                                  None) == -3


_ENGINE_LEG = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from clsurvey_amd import models
from clsurvey_amd.net import NetEngine
torch.manual_seed(3)
m = models.parse_model_name("alexnet_scratch", num_classes=40)
N = 6
eng = NetEngine(m, N, (3, 224, 224), "cuda")
eng.auto_dropout = False
g = torch.Generator().manual_seed(5)
x = torch.randn(N, 3, 224, 224, generator=g).cuda(); y = torch.randint(0, 40, (N,), generator=g).cuda()
m.eval()
loss, logits = eng.loss_step(x, y, "ce_mean", True, want_logits=True)
torch.cuda.synchronize()
np.savez(sys.argv[1], loss=loss.cpu().numpy(), logits=logits.cpu().numpy(), grad=eng.arena.grad.cpu().numpy())
"""


def test_plan_executor_takes_the_path_for_alexnet_conv1(tmp_path):
    """Same model, same batch, two processes: CLHIP_S2D=1 (space-to-depth first layer) and the default (gather-GEMM): loss, logits
    and the whole gradient arena agree to fp32 summation order (1e-4 of each quantity's scale; the first layer's own dW among them)."""
    outs = []
    for tag, env in (("s2d", {"CLHIP_S2D": "1"}), ("gemm", {"CLHIP_S2D": "0"})):
        out = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, "-c", _ENGINE_LEG % ROOT, out], env=dict(os.environ, **env), capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    a, b = outs
    assert abs(float(a["loss"].reshape(-1)[0]) - float(b["loss"].reshape(-1)[0])) <= 1e-5 * abs(float(b["loss"].reshape(-1)[0]))
    assert np.abs(a["logits"] - b["logits"]).max() <= 1e-4 * np.abs(b["logits"]).max()
    assert np.abs(a["grad"] - b["grad"]).max() <= 1e-4 * np.abs(b["grad"]).max()
    n1 = 64 * 3 * 11 * 11                 # conv1's weights lead the arena (models/net.py: features.0)
    assert np.abs(a["grad"][:n1] - b["grad"][:n1]).max() <= 1e-4 * np.abs(b["grad"][:n1]).max()
    assert not np.array_equal(a["grad"][:n1], b["grad"][:n1]), "both legs ran the same kernel: the switch did nothing"
