"""Backward of a small 3x3 layer as ONE grid (csrc/wino.hip, wino_pair_kernel; clhip_conv3x3_wino_bwd): backward-data and weight
gradient blocks of the layer interleaved in one launch.  The two halves are the device code of the two kernels of their own, so the
results must be BIT-identical to clhip_conv3x3_wino_bwd_data + clhip_conv3x3_wino_bwd_weight — and through them equal to the
autograd backward of the reference's conv layers (VGGSlim.py:27-40), which tests/test_gpu_wino.py holds those two to."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch_ref  # noqa: E402

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _layer(N, C, K, H, W, pooled, seed):
    from clsurvey_amd import ops
    g = torch.Generator(device=_dev())
    g.manual_seed(seed)
    x = torch.randn((N, C, H, W), generator=g, device=_dev())
    w = torch.randn((K, C, 3, 3), generator=g, device=_dev()) * 0.05
    b = torch.randn((K,), generator=g, device=_dev()) * 0.1
    if pooled:
        yp, idx = ops.conv3x3_wino_fwd(x, w, b, True, pool=True)
        dy = torch.randn(yp.shape, generator=g, device=_dev())
        return x, w, dy, idx
    dy = torch.randn((N, K, H, W), generator=g, device=_dev())
    return x, w, dy, None


# the deep layers of the VGG9s at the bench batch, ragged batches (odd N: half-filled blocks on both halves, unequal block counts so
# that the tail of the longer list runs alone), non-square maps, more channels
SHAPES = [(200, 64, 64, 16, 16, False), (200, 64, 64, 16, 16, True), (200, 64, 128, 8, 8, False), (200, 128, 128, 8, 8, True),
          (13, 64, 64, 16, 16, True), (3, 128, 64, 8, 8, False), (37, 64, 192, 8, 16, False), (21, 128, 64, 12, 16, True),
          (50, 256, 128, 8, 8, True), (1, 64, 64, 16, 16, False),
          # base_VGG9's deep layers at the bench batch (BASELINE config 3: its four deep layers take the merged grid)
          (200, 64, 128, 16, 16, False), (200, 128, 128, 16, 16, True), (200, 128, 256, 8, 8, False), (200, 256, 256, 8, 8, True)]


@pytest.mark.parametrize("N,C,K,H,W,pooled", SHAPES)
def test_one_grid_backward_is_bit_identical_to_the_two_launches(N, C, K, H, W, pooled):
    from clsurvey_amd import ops
    x, w, dy, idx = _layer(N, C, K, H, W, pooled, 11 + N)
    for mask in ((None,) if pooled else (None, x)):      # behind a fused pool the plan passes no ReLU source (dead windows are in the codes)
        got = ops.conv3x3_wino_bwd(x, dy, w, mask, idx)
        if got is None and C >= 256:
            pytest.skip("from 256 input channels on the weight gradient runs on the 64 x 64-tile kernel: no merged grid (the plan "
                        "executor issues the two launches, tests/test_gpu_wino.py)")
        assert got is not None, "the merged grid should take this layer"
        dx = ops.conv3x3_wino_bwd_data(dy, w, mask, idx)
        dw, db = ops.conv3x3_wino_bwd_weight(x, dy, idx)
        assert torch.equal(got[0], dx)
        assert torch.equal(got[1], dw) and torch.equal(got[2], db)
        again = ops.conv3x3_wino_bwd(x, dy, w, mask, idx)          # deterministic (slabs reduced in a fixed order)
        assert all(torch.equal(a, b) for a, b in zip(got, again))


def test_one_grid_backward_against_autograd():
    """... and against torch's autograd in fp64 directly (one pooled, one plain layer), at the tolerance of test_gpu_wino.py.  The pooled
    gradient is un-pooled with the op's own arg-max codes (0..3 = window position, 4 = dead window), so that a near-tie between the fp32
    forward and an fp64 one cannot move a window."""
    import torch.nn.functional as F
    from clsurvey_amd import ops
    for (N, C, K, H, W, pooled) in ((24, 64, 64, 16, 16, True), (24, 64, 128, 8, 8, False)):
        x, w, dy, idx = _layer(N, C, K, H, W, pooled, 5)
        xd = x.double().requires_grad_(True)
        wd = w.double().requires_grad_(True)
        bd = torch.zeros(K, dtype=torch.float64, device=_dev(), requires_grad=True)
        z = F.conv2d(xd, wd, bd, padding=1)
        if pooled:
            up = torch.zeros((N, K, H, W), dtype=torch.float64, device=_dev())
            for code in range(4):
                up[:, :, (code >> 1)::2, (code & 1)::2] = torch.where(idx == code, dy.double(), torch.zeros_like(dy, dtype=torch.float64))
            z.backward(up)
        else:
            z.backward(dy.double())
        got = ops.conv3x3_wino_bwd(x, dy, w, None, idx)
        assert got is not None
        for name, a, ref in (("dx", got[0], xd.grad), ("dw", got[1], wd.grad), ("db", got[2], bd.grad)):
            err = (a.double() - ref).abs().max().item()
            assert err <= 2e-5 * ref.abs().max().item(), (name, err, ref.abs().max().item())


@pytest.mark.parametrize("N,C,K,H,W,pooled", [(200, 64, 64, 16, 16, True), (200, 64, 128, 8, 8, False), (200, 128, 128, 8, 8, True),
                                              (200, 64, 128, 16, 16, False), (200, 128, 128, 16, 16, True), (200, 128, 256, 8, 8, False)])
def test_one_grid_backward_against_torch_cpu_at_the_bench_batch(N, C, K, H, W, pooled):
    """The merged grid's instances depend on N (block counts, narrow / wide wave tiles): small_VGG9's and base_VGG9's deep layers at
    batch 200 against torch CPU's fp32 autograd (conv_transpose2d for dx, conv2d backward for dW / db), the pooled gradient un-pooled
    with the op's own codes."""
    import torch.nn.functional as F
    from clsurvey_amd import ops
    x, w, dy, idx = _layer(N, C, K, H, W, pooled, 23)
    got = ops.conv3x3_wino_bwd(x, dy, w, None, idx)
    assert got is not None, "the merged grid should take this layer"
    up = torch_ref.unpool(dy, idx) if pooled else dy.cpu()
    dx_ref = F.conv_transpose2d(up, w.cpu(), padding=1)
    dw_ref, db_ref = torch_ref.bwd_weight(x, up)
    for name, a, ref, tol in (("dx", got[0], dx_ref, 2e-5), ("dw", got[1], dw_ref, 5e-5), ("db", got[2], db_ref, 5e-5)):
        err = (a.cpu().double() - ref.double()).abs().max().item()
        assert err <= tol * ref.abs().max().item(), (name, err, ref.abs().max().item())


@pytest.mark.parametrize("N,C,K,H,W,pooled", [(200, 64, 64, 32, 32, True), (8, 64, 64, 12, 12, False), (8, 64, 64, 13, 13, False),
                                              (8, 64, 64, 32, 16, False), (4, 48, 64, 16, 16, False)])
def test_one_grid_backward_declines_other_shapes(N, C, K, H, W, pooled):
    """Maps of more than 256 pixels (their launches fill the chip on their own), maps narrower than 16 other than 8 x 8, odd maps,
    channel counts outside the Winograd weight gradient: CLHIP_ENOTSUP, nothing launched (dx / dw untouched)."""
    from clsurvey_amd import ops, _lib
    x = torch.randn((N, C, H, W), device=_dev())
    w = torch.randn((K, C, 3, 3), device=_dev())
    dy = torch.randn((N, K, H // 2, W // 2) if pooled else (N, K, H, W), device=_dev())
    idx = torch.zeros((N, K, H // 2, W // 2), dtype=torch.uint8, device=_dev()) if pooled else None
    assert ops.conv3x3_wino_bwd(x, dy, w, None, idx) is None
    L = _lib.lib()
    assert L.clhip_conv3x3_wino_bwd(x.data_ptr(), dy.data_ptr(), None, w.data_ptr(), None, x.data_ptr(), w.data_ptr(), None, N, C, K, H, W,
                                    None, 0, None) in (-1, -3)          # no workspace: refused before any launch
