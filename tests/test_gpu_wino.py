"""Winograd F(2x2, 3x3) convolution kernels (csrc/wino.hip) against torch CPU (the reference's arithmetic: nn.Conv2d + ReLU +
MaxPool2d of models/VGGSlim.py:27-40 and their autograd backward) and against the direct MFMA kernels."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch_ref  # noqa: E402

pytestmark = pytest.mark.gpu

SHAPES = [  # N, C, K, H, W
    (3, 64, 64, 32, 32), (5, 64, 64, 16, 16), (6, 64, 128, 8, 8), (9, 128, 128, 8, 8), (2, 16, 32, 8, 8),
    (2, 24, 96, 12, 20), (3, 128, 256, 16, 16), (2, 64, 64, 64, 64), (7, 256, 256, 8, 8), (2, 32, 64, 28, 28),
    (200, 64, 64, 32, 32), (200, 64, 64, 16, 16), (200, 128, 128, 8, 8),
    (131, 128, 128, 8, 8), (200, 64, 128, 8, 8),        # 8 x 8 maps with few units: the 16x16x4-MFMA variant, both wave shapes, odd image count
    # every 3x3 layer shape base_VGG9 / wide_VGG9 run at the bench batch (BASELINE configs 3 and 5; dispatch depends on N)
    (200, 64, 128, 16, 16), (200, 128, 128, 16, 16), (200, 128, 256, 8, 8), (200, 256, 256, 8, 8),
    (200, 64, 128, 32, 32), (200, 128, 256, 16, 16), (200, 256, 256, 16, 16), (200, 256, 512, 8, 8), (200, 512, 512, 8, 8),
    # odd maps (AlexNet's 13 x 13 layers, models/net.py:96-125): row-packed geometry, half-outside last tile row / column
    (4, 32, 64, 13, 13), (3, 64, 32, 13, 13), (2, 16, 32, 9, 15), (5, 64, 64, 11, 13), (3, 32, 32, 13, 16), (37, 192, 384, 13, 13),
]


def _rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / max(float(b.double().abs().max()), 1e-30))


def _check_fused_pool(ops, xd, wd, bd, y_ref, N, K, H, W, big):
    # fused ReLU + 2x2 max-pool: value within rounding; the arg-max code must name an element that attains the maximum
    yp, idx = ops.conv3x3_wino_fwd(xd, wd, bd, relu=True, pool=True)
    yc = y_ref.cuda() if not big else y_ref
    win = yc.reshape(N, K, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(N, K, H // 2, W // 2, 4)
    assert _rel(yp, win.max(4).values) <= 2e-5
    # (code 4 = the window has no positive maximum after ReLU: ReLU folded into the code, csrc/common.hpp)
    assert torch.equal(idx == 4, yp == 0)
    live = idx.long().where(idx < 4, torch.zeros((), dtype=torch.long, device=idx.device))
    picked = torch.gather(win, 4, live.unsqueeze(-1)).squeeze(-1)
    assert float((picked - win.max(4).values).abs().max()) <= 2e-5 * float(yc.abs().max())
    assert int(idx.max()) <= 4
    ypn, idxn = ops.conv3x3_wino_fwd(xd, wd, bd, relu=False, pool=True)           # without ReLU the codes are plain arg-max bytes
    assert int(idxn.max()) <= 3


@pytest.mark.parametrize("shape", SHAPES)
def test_wino_forward_and_backward_data(shape):
    from clsurvey_amd import ops
    N, C, K, H, W = shape
    gen = np.random.RandomState(N * 7 + C + K + H)
    x = torch.from_numpy(gen.standard_normal((N, C, H, W)).astype(np.float32))
    w = torch.from_numpy((gen.standard_normal((K, C, 3, 3)) * (2.0 / (9 * C)) ** 0.5).astype(np.float32))
    b = torch.from_numpy((gen.standard_normal((K,)) * 0.1).astype(np.float32))
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    big = N >= 100
    # judge: torch CPU at every batch size (tests/torch_ref.py)
    z_ref = F.conv2d(x, w, b, padding=1)
    y_ref = F.relu(z_ref)
    if big:
        y_ref = y_ref.cuda()
    assert _rel(ops.conv3x3_wino_fwd(xd, wd, bd, relu=False), z_ref) <= 2e-5
    y = ops.conv3x3_wino_fwd(xd, wd, bd, relu=True)
    assert _rel(y, y_ref) <= 2e-5
    odd = bool((H | W) & 1)
    if not odd:
        _check_fused_pool(ops, xd, wd, bd, y_ref, N, K, H, W, big)
    if C % 32 or K < 16:
        return              # backward-data runs the same kernel with the channel roles swapped: its own shape domain
    # backward-data (+ ReLU mask of the producing layer), plain and from the pooled gradient
    dy = torch.from_numpy(gen.standard_normal((N, K, H, W)).astype(np.float32))
    msrc = torch.from_numpy(gen.standard_normal((N, C, H, W)).astype(np.float32))
    dyd, md = dy.cuda(), msrc.cuda()
    dx_ref = F.conv_transpose2d(dy, w, padding=1)
    assert _rel(ops.conv3x3_wino_bwd_data(dyd, wd), dx_ref) <= 2e-5
    dxm = ops.conv3x3_wino_bwd_data(dyd, wd, relu_src=md)
    assert _rel(dxm, dx_ref.cuda() * (md > 0)) <= 2e-5
    if odd:
        return              # no 2x2 pooling on odd maps
    dyp = torch.from_numpy(gen.standard_normal((N, K, H // 2, W // 2)).astype(np.float32)).cuda()
    code = torch.from_numpy(gen.randint(0, 5, size=(N, K, H // 2, W // 2)).astype(np.uint8)).cuda()
    dy_full = torch_ref.unpool(dyp, code)
    ref_u = F.conv_transpose2d(dy_full, w, padding=1) * (msrc > 0)
    assert _rel(ops.conv3x3_wino_bwd_data(dyp, wd, relu_src=md, idx=code), ref_u) <= 2e-5


def test_wino_refuses_shapes_outside_its_domain():
    from clsurvey_amd import ops, _lib
    x = torch.zeros(1, 3, 8, 8, device="cuda")
    with pytest.raises(_lib.ClhipError):
        ops.conv3x3_wino_fwd(x, torch.zeros(32, 3, 3, 3, device="cuda"), torch.zeros(32, device="cuda"))
    x = torch.zeros(1, 16, 21, 21, device="cuda")         # odd maps wider than 16 (9..16 go through the row-packed geometry)
    with pytest.raises(_lib.ClhipError):
        ops.conv3x3_wino_fwd(x, torch.zeros(32, 16, 3, 3, device="cuda"), torch.zeros(32, device="cuda"))


WG_SHAPES = [(3, 64, 64, 32, 32), (5, 64, 64, 16, 16), (6, 64, 128, 8, 8), (4, 128, 64, 8, 8), (2, 64, 64, 12, 20), (2, 64, 128, 28, 28),
             (3, 128, 256, 16, 16), (200, 64, 64, 32, 32), (200, 64, 64, 16, 16), (200, 128, 128, 8, 8),
             (200, 64, 128, 16, 16), (200, 128, 128, 16, 16), (200, 128, 256, 8, 8), (200, 256, 256, 8, 8),      # base_VGG9 at the bench batch
             (200, 64, 128, 32, 32), (200, 128, 256, 16, 16), (200, 256, 256, 16, 16), (200, 256, 512, 8, 8), (200, 512, 512, 8, 8),   # wide_VGG9
             (6, 64, 64, 13, 13), (3, 128, 64, 9, 11), (4, 64, 128, 15, 16), (40, 192, 384, 13, 13)]


@pytest.mark.parametrize("shape", WG_SHAPES)
def test_wino_weight_gradient(shape):
    """dW, db through G^T[(A dY A^T).*(B^T d B)]G against torch CPU autograd (every shape, the bench batch included),
    plain and from the POOLED gradient + arg-max codes; repeated calls are bitwise equal (fixed-order slab reduction)."""
    from clsurvey_amd import ops
    N, C, K, H, W = shape
    gen = np.random.RandomState(N * 11 + C + K + W)
    x = torch.from_numpy(gen.standard_normal((N, C, H, W)).astype(np.float32))
    dy = torch.from_numpy(gen.standard_normal((N, K, H, W)).astype(np.float32))
    xd, dyd = x.cuda(), dy.cuda()
    dw_ref, db_ref = torch_ref.bwd_weight(x, dy)             # torch CPU autograd at every batch size
    dw, db = ops.conv3x3_wino_bwd_weight(xd, dyd)
    assert _rel(dw, dw_ref) <= 5e-5 and _rel(db, db_ref) <= 5e-5, (_rel(dw, dw_ref), _rel(db, db_ref))
    dw2, db2 = ops.conv3x3_wino_bwd_weight(xd, dyd)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    if (H | W) & 1:
        return              # no 2x2 pooling on odd maps
    dyp = torch.from_numpy(gen.standard_normal((N, K, H // 2, W // 2)).astype(np.float32)).cuda()
    code = torch.from_numpy(gen.randint(0, 5, size=(N, K, H // 2, W // 2)).astype(np.uint8)).cuda()
    dw_u_ref, db_u_ref = torch_ref.bwd_weight(x, torch_ref.unpool(dyp, code))
    dw_u, db_u = ops.conv3x3_wino_bwd_weight(xd, dyp, idx=code)
    assert _rel(dw_u, dw_u_ref) <= 5e-5 and _rel(db_u, db_u_ref) <= 5e-5


@pytest.mark.parametrize("C,K", [(24, 96), (64, 64), (128, 32)])
def test_wino_weight_images_agree(C, K):
    """The transformed weights U = G g G^T are kept twice: the LDS image [k-tile][8-channel chunk][channel][out channel]
    [16 frequencies + 4 pad] and, behind it, the lane-ordered image wino_conv16g_kernel reads straight from L2 into its A-operand
    registers: [k-tile][4-channel chunk][out-channel half][row tile][frequency quad][lane = 16 * channel + (out channel & 15)][4].
    Both must hold the same bits, the LDS image must be G g G^T (Lavin & Gray's F(2x2, 3x3) weight transform), padding is zero."""
    from clsurvey_amd import _lib, ops
    L = _lib.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, C, 8, 8, generator=g).to(dev)
    w = torch.randn(K, C, 3, 3, generator=g) * 0.1
    wd, b = w.to(dev), torch.zeros(K, device=dev)
    y = torch.empty(2, K, 8, 8, device=dev)
    ws = torch.zeros(L.clhip_conv3x3_wino_ws(C, K), dtype=torch.uint8, device=dev)
    ops.check(L.clhip_conv3x3_wino_fwd(ops._ptr(x), ops._ptr(wd), ops._ptr(b), ops._ptr(y), None, 2, C, K, 8, 8, 0, ops._ptr(ws),
                                       ws.numel(), ops._stream()), "clhip_conv3x3_wino_fwd")
    torch.cuda.synchronize()
    U = ws.view(torch.float32).cpu().numpy()
    kts, nch = (K + 63) // 64, (C + 7) // 8
    n_lds = kts * nch * 8 * 64 * 20
    lds = U[:n_lds].reshape(kts, nch, 8, 64, 20)
    assert np.all(lds[..., 16:] == 0.0)
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
    want = np.zeros((kts * 64, nch * 8, 16))
    want[:K, :C] = np.einsum("ar,kcrs,bs->kcab", G, w.double().numpy(), G).reshape(K, C, 16)
    got = lds[..., :16].transpose(0, 3, 1, 2, 4).reshape(kts * 64, nch * 8, 16)          # [k][c][f]
    assert np.abs(got - want).max() <= 1e-6 * max(1.0, np.abs(want).max())
    if U.size < n_lds + kts * nch * 2 * 4096:
        pytest.skip("library built with CLHIP_W16G_ADIRECT=0: no lane-ordered image")
    direct = U[n_lds:n_lds + kts * nch * 2 * 4096].reshape(kts, 2 * nch, 2, 2, 4, 4, 16, 4)      # kt, chunk4, wk, r, fq, q, ti, j
    # the same elements out of the LDS image: channel = 8 chunk8 + 4 half + q, out channel = 32 wk + 16 r + ti, frequency = 4 fq + j
    e = lds[..., :16].reshape(kts, nch, 2, 4, 2, 2, 16, 4, 4)                                  # kt, chunk8, half, q, wk, r, ti, fq, j
    e = e.transpose(0, 1, 2, 4, 5, 7, 3, 6, 8).reshape(kts, 2 * nch, 2, 2, 4, 4, 16, 4)
    assert np.array_equal(direct.view(np.uint32), np.ascontiguousarray(e).view(np.uint32))
