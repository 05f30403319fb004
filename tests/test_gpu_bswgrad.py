"""Weight gradient of the 3x3 layers on the bf16 matrix cores with split fp32 operands (csrc/bswgrad.hip) against torch CPU autograd
(convolution_backward of nn.Conv2d, models/VGGSlim.py:27-40): plain and from the POOLED gradient + arg-max codes, at the bench batch of
every VGG9 width, ragged batches and maps; bitwise determinism; and its error against fp64 next to the f32 kernels'."""
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
    (3, 64, 64, 32, 32), (5, 64, 64, 16, 16), (2, 64, 128, 20, 16), (7, 128, 64, 9, 32), (1, 64, 64, 1, 16), (2, 64, 64, 64, 64),
    (3, 128, 256, 16, 16), (13, 64, 64, 30, 48), (2, 256, 64, 17, 16),
    (200, 64, 64, 32, 32), (200, 64, 64, 16, 16), (131, 64, 64, 32, 32),
    (200, 64, 128, 16, 16), (200, 128, 128, 16, 16), (200, 64, 128, 32, 32), (200, 128, 256, 16, 16), (200, 256, 256, 16, 16),
    # maps that are not whole 16-pixel strips (AlexNet's 13 x 13 layers, models/net.py:96-125): the tap-split kernel of bswgrad5.hip
    (37, 192, 384, 13, 13), (4, 32, 64, 13, 13), (3, 64, 32, 9, 15), (5, 64, 64, 11, 13), (2, 32, 32, 8, 8), (3, 32, 96, 12, 20), (128, 256, 256, 13, 13),
]


def _rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / max(float(b.double().abs().max()), 1e-30))


@pytest.mark.parametrize("shape", SHAPES)
def test_bs_weight_gradient(shape):
    from clsurvey_amd import ops
    N, C, K, H, W = shape
    gen = np.random.RandomState(N * 11 + C + K + W)
    x = torch.from_numpy(gen.standard_normal((N, C, H, W)).astype(np.float32))
    dy = torch.from_numpy(gen.standard_normal((N, K, H, W)).astype(np.float32))
    xd, dyd = x.cuda(), dy.cuda()
    dw_ref, db_ref = torch_ref.bwd_weight(x, dy)
    dw, db = ops.conv3x3_bs_bwd_weight(xd, dyd)
    assert _rel(dw, dw_ref) <= 5e-5 and _rel(db, db_ref) <= 5e-5, (_rel(dw, dw_ref), _rel(db, db_ref))
    dw2, db2 = ops.conv3x3_bs_bwd_weight(xd, dyd)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    if ((H | W) & 1) or W % 16 or C % 64 or K % 64:
        return              # (pooled gradients: the 16-pixel-aligned kernel only)
    dyp = torch.from_numpy(gen.standard_normal((N, K, H // 2, W // 2)).astype(np.float32)).cuda()
    code = torch.from_numpy(gen.randint(0, 5, size=(N, K, H // 2, W // 2)).astype(np.uint8)).cuda()
    dw_u_ref, db_u_ref = torch_ref.bwd_weight(x, torch_ref.unpool(dyp, code))
    dw_u, db_u = ops.conv3x3_bs_bwd_weight(xd, dyp, idx=code)
    assert _rel(dw_u, dw_u_ref) <= 5e-5 and _rel(db_u, db_u_ref) <= 5e-5, (_rel(dw_u, dw_u_ref), _rel(db_u, db_u_ref))
    # the un-pooled row is rebuilt in registers: bitwise the kernel on the un-pooled gradient
    dw_f, db_f = ops.conv3x3_bs_bwd_weight(xd, ops.maxpool2_bwd(dyp, code))
    assert torch.equal(dw_u, dw_f) and torch.equal(db_u, db_f)


def test_bs_weight_gradient_refuses_shapes_outside_its_domain():
    from clsurvey_amd import _lib
    L = _lib.lib()
    assert L.clhip_conv3x3_bs_bwd_weight_ws(8, 64, 64, 32, 32) > 0
    for C, K, H, W in ((3, 64, 64, 64), (48, 64, 16, 16), (64, 80, 16, 16), (64, 64, 4, 4)):
        assert L.clhip_conv3x3_bs_bwd_weight_ws(8, C, K, H, W) == 0
    x = torch.zeros(2, 64, 4, 4, device="cuda")
    assert L.clhip_conv3x3_bs_bwd_weight(x.data_ptr(), x.data_ptr(), None, x.data_ptr(), x.data_ptr(), 2, 64, 64, 4, 4, x.data_ptr(), 1 << 20,
                                         None) == -3


@pytest.mark.parametrize("C,K,N,HW", [(64, 64, 16, 32), (128, 128, 8, 16)])
def test_bs_weight_gradient_error_is_that_of_an_fp32_chain(C, K, N, HW):
    """Against an fp64 weight gradient on the same fp32 inputs, in units of sum |dy x| per element: the split path's error must not
    exceed the Winograd f32 kernel's nor the direct f32 MFMA kernel's on the same data (the bench line's `dtype: f32` rests on it)."""
    from clsurvey_amd import ops
    gen = np.random.RandomState(C + K + N)
    x = torch.from_numpy(np.maximum(gen.standard_normal((N, C, HW, HW)), 0).astype(np.float32))     # a ReLU output
    dy = torch.from_numpy(gen.standard_normal((N, K, HW, HW)).astype(np.float32))
    ref, _ = torch_ref.bwd_weight(x.double(), dy.double())
    scale, _ = torch_ref.bwd_weight(x.double().abs(), dy.double().abs())
    scale = scale.clamp_min(1e-30)
    xd, dyd = x.cuda(), dy.cuda()

    def err(dw):
        e = (dw.double().cpu() - ref).abs() / scale
        return float(e.max()), float((e * e).mean().sqrt())
    bs, wino, direct = err(ops.conv3x3_bs_bwd_weight(xd, dyd)[0]), err(ops.conv3x3_wino_bwd_weight(xd, dyd)[0]), err(ops.conv3x3_bwd_weight(xd, dyd)[0])
    print("C=%d K=%d N=%d @%d  error / sum|dy x| (max, rms): bf16-split %.3e %.3e   Winograd f32 %.3e %.3e   direct f32 MFMA %.3e %.3e"
          % ((C, K, N, HW) + bs + wino + direct))
    # (sums of 16 k - 33 k products with random signs: all three sit far below one fp32 ulp of the sum of magnitudes)
    assert bs[0] <= 2.5e-7 and bs[1] <= 5e-8
    assert bs[1] <= 2.0 * max(wino[1], direct[1]) and bs[0] <= 2.0 * max(wino[0], direct[0])


SHAPES5 = [  # N, C, K, H, W — 5x5 / padding-2 weight gradient (csrc/bswgrad5.hip); AlexNet's second convolution first
    (128, 64, 192, 27, 27), (3, 32, 64, 16, 16), (2, 64, 32, 9, 20), (5, 32, 32, 27, 27), (2, 32, 64, 5, 48), (1, 32, 32, 1, 16), (7, 64, 64, 31, 17),
]


@pytest.mark.parametrize("shape", SHAPES5)
def test_bs_weight_gradient_5x5(shape):
    """dW, db of nn.Conv2d(C, K, 5, padding=2) (models/net.py:96-125) against torch CPU autograd and against the gather-GEMM entry point
    of the same operator; bitwise deterministic."""
    from clsurvey_amd import ops
    N, C, K, H, W = shape
    gen = np.random.RandomState(N * 13 + C + K + W)
    x = torch.from_numpy(gen.standard_normal((N, C, H, W)).astype(np.float32))
    dy = torch.from_numpy(gen.standard_normal((N, K, H, W)).astype(np.float32))
    xd, dyd = x.cuda(), dy.cuda()
    dw_ref, db_ref = torch_ref.bwd_weight(x, dy, ksize=5)
    dw, db = ops.conv5x5_bs_bwd_weight(xd, dyd)
    assert _rel(dw, dw_ref) <= 5e-5 and _rel(db, db_ref) <= 5e-5, (_rel(dw, dw_ref), _rel(db, db_ref))
    dw2, db2 = ops.conv5x5_bs_bwd_weight(xd, dyd)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    dwg, dbg = ops.conv2d_bwd_weight(xd, dyd, (5, 5), 1, 2)
    assert _rel(dwg, dw) <= 5e-5 and _rel(dbg, db) <= 5e-5


def test_bs_weight_gradient_5x5_refuses_shapes_outside_its_domain():
    from clsurvey_amd import _lib
    L = _lib.lib()
    assert L.clhip_conv5x5_bs_bwd_weight_ws(8, 64, 192, 27, 27) > 0
    for C, K, H, W in ((3, 64, 27, 27), (48, 64, 27, 27), (64, 80, 27, 27), (64, 64, 4, 4)):
        assert L.clhip_conv5x5_bs_bwd_weight_ws(8, C, K, H, W) == 0
