"""3x3 convolution on the bf16 matrix cores with split fp32 operands (csrc/bsconv.hip) against torch CPU (the reference's arithmetic:
nn.Conv2d + ReLU + MaxPool2d of models/VGGSlim.py:27-40 and their autograd backward w.r.t. the input), against the direct f32 MFMA
kernels at the bench sizes, and against an fp64 convolution: the split path's error must stay at the level of an fp32 chain."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch_ref  # noqa: E402

SHAPES = [  # N, C, K, H, W
    (3, 64, 64, 32, 32), (5, 64, 64, 16, 16), (6, 64, 128, 8, 8), (9, 128, 128, 8, 8), (2, 32, 64, 8, 8),
    (2, 32, 64, 12, 20), (3, 128, 256, 16, 16), (2, 64, 64, 64, 64), (7, 256, 256, 8, 8), (2, 32, 64, 28, 28),
    (1, 64, 64, 4, 4), (3, 64, 64, 6, 10), (2, 64, 192, 56, 56),                      # maps smaller / other than the block regions
    (200, 64, 64, 32, 32), (200, 64, 64, 16, 16), (200, 128, 128, 8, 8), (131, 128, 128, 8, 8), (200, 64, 128, 8, 8),
    # every 3x3 layer shape base_VGG9 / wide_VGG9 run at the bench batch (BASELINE configs 3 and 5; dispatch depends on N)
    (200, 64, 128, 16, 16), (200, 128, 128, 16, 16), (200, 128, 256, 8, 8), (200, 256, 256, 8, 8),
    (200, 64, 128, 32, 32), (200, 128, 256, 16, 16), (200, 256, 256, 16, 16), (200, 256, 512, 8, 8), (200, 512, 512, 8, 8),
    # odd maps (AlexNet's 13 x 13 layers, models/net.py:96-125): tiles past the edge, 4-byte stores, no fused pooling
    (4, 32, 64, 13, 13), (3, 64, 64, 13, 13), (2, 32, 64, 9, 15), (5, 64, 64, 11, 13), (37, 192, 384, 13, 13),
]


def _rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / max(float(b.double().abs().max()), 1e-30))


def _check_fused_pool(ops, xd, wd, bd, y_ref, N, K, H, W, big):
    # fused ReLU + 2x2 max-pool: value within rounding; the arg-max code must name an element that attains the maximum
    yp, idx = ops.conv3x3_bs_fwd(xd, wd, bd, relu=True, pool=True)
    yc = y_ref.cuda() if not big else y_ref
    win = yc.reshape(N, K, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(N, K, H // 2, W // 2, 4)
    assert _rel(yp, win.max(4).values) <= 2e-5
    assert torch.equal(idx == 4, yp == 0)           # code 4 = no positive maximum after ReLU (csrc/common.hpp)
    live = idx.long().where(idx < 4, torch.zeros((), dtype=torch.long, device=idx.device))
    picked = torch.gather(win, 4, live.unsqueeze(-1)).squeeze(-1)
    assert float((picked - win.max(4).values).abs().max()) <= 2e-5 * float(yc.abs().max())
    assert int(idx.max()) <= 4
    ypn, idxn = ops.conv3x3_bs_fwd(xd, wd, bd, relu=False, pool=True)             # without ReLU the codes are plain arg-max bytes
    assert int(idxn.max()) <= 3
    # and the codes are the FIRST maximum in ATen's scan order, judged on the kernel's own un-pooled output (bitwise the same sums)
    z = ops.conv3x3_bs_fwd(xd, wd, bd, relu=False)
    zw = z.reshape(N, K, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(N, K, H // 2, W // 2, 4)
    assert torch.equal(idxn.long(), zw.argmax(4)) or float((zw.max(4).values - torch.gather(zw, 4, idxn.long().unsqueeze(-1)).squeeze(-1)).abs().max()) == 0.0
    assert torch.equal(ypn, zw.max(4).values)


@pytest.mark.parametrize("shape", SHAPES)
def test_bs_forward_and_backward_data(shape):
    _run((shape,))


def _run(shapes):
    from clsurvey_amd import ops
    for shape in shapes:
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
        assert _rel(ops.conv3x3_bs_fwd(xd, wd, bd, relu=False), z_ref) <= 2e-5, shape
        y = ops.conv3x3_bs_fwd(xd, wd, bd, relu=True)
        assert _rel(y, y_ref) <= 2e-5, shape
        odd = bool((H | W) & 1)
        if not odd:
            _check_fused_pool(ops, xd, wd, bd, y_ref, N, K, H, W, big)
        if C % 64 or K % 32:
            continue            # backward-data runs the same kernel with the channel roles swapped: its own shape domain
        dy = torch.from_numpy(gen.standard_normal((N, K, H, W)).astype(np.float32))
        msrc = torch.from_numpy(gen.standard_normal((N, C, H, W)).astype(np.float32))
        dyd, md = dy.cuda(), msrc.cuda()
        dx_ref = F.conv_transpose2d(dy, w, padding=1)
        assert _rel(ops.conv3x3_bs_bwd_data(dyd, wd), dx_ref) <= 2e-5, shape
        dxm = ops.conv3x3_bs_bwd_data(dyd, wd, relu_src=md)
        assert _rel(dxm, dx_ref.cuda() * (md > 0)) <= 2e-5, shape
        if odd:
            continue            # no 2x2 pooling on odd maps
        dyp = torch.from_numpy(gen.standard_normal((N, K, H // 2, W // 2)).astype(np.float32)).cuda()
        code = torch.from_numpy(gen.randint(0, 5, size=(N, K, H // 2, W // 2)).astype(np.uint8)).cuda()
        dy_full = ops.maxpool2_bwd(dyp, code)
        assert torch.equal(dy_full.cpu(), torch_ref.unpool(dyp, code))
        ref_u = F.conv_transpose2d(dy_full.cpu(), w, padding=1) * (msrc > 0)
        assert _rel(ops.conv3x3_bs_bwd_data(dyp, wd, relu_src=md, idx=code), ref_u) <= 2e-5, shape
        # the un-pooling staging feeds the same sums as the plain one: bitwise equal to the kernel on the un-pooled gradient
        assert torch.equal(ops.conv3x3_bs_bwd_data(dyp, wd, relu_src=md, idx=code), ops.conv3x3_bs_bwd_data(dy_full, wd, relu_src=md))


@pytest.mark.parametrize("N,C,K,HW", [(200, 64, 64, 32), (100, 64, 64, 32), (200, 64, 128, 32), (50, 256, 512, 28)])
def test_two_geometry_launch_is_bitwise(N, C, K, HW):
    """Launches with whole rounds of 128-pixel blocks plus a short remainder put the last images out as 64-pixel tiles in the same grid
    (bs_conv_mixed_kernel).  An output does not depend on the tile it falls in: every image equals, bit for bit, the same entry point
    run on a batch that takes the one-geometry launch (the first 8 images; the last 8; a middle chunk)."""
    from clsurvey_amd import ops
    g = torch.Generator(device="cuda")
    g.manual_seed(N + C + K)
    x = torch.randn((N, C, HW, HW), generator=g, device="cuda").relu_()
    w = torch.randn((K, C, 3, 3), generator=g, device="cuda") * (2.0 / (9 * C)) ** 0.5
    b = torch.randn((K,), generator=g, device="cuda") * 0.1
    dy = torch.randn((N, K, HW, HW), generator=g, device="cuda")
    dyp = torch.randn((N, K, HW // 2, HW // 2), generator=g, device="cuda")
    code = torch.randint(0, 5, (N, K, HW // 2, HW // 2), generator=g, device="cuda", dtype=torch.uint8)
    chunks = [(0, 8), (N - 8, N), (N // 2 - 4, N // 2 + 4)]
    yp, idx = ops.conv3x3_bs_fwd(x, w, b, relu=True, pool=True)
    y = ops.conv3x3_bs_fwd(x, w, b, relu=True)
    if C % 64 == 0 and K % 32 == 0:
        dx = ops.conv3x3_bs_bwd_data(dy, w, relu_src=x)
        dxu = ops.conv3x3_bs_bwd_data(dyp, w, relu_src=None, idx=code)
    for a, e in chunks:
        xs = x[a:e].contiguous()
        yps, idxs = ops.conv3x3_bs_fwd(xs, w, b, relu=True, pool=True)
        assert torch.equal(yp[a:e], yps) and torch.equal(idx[a:e], idxs), (a, e)
        assert torch.equal(y[a:e], ops.conv3x3_bs_fwd(xs, w, b, relu=True)), (a, e)
        if C % 64 == 0 and K % 32 == 0:
            assert torch.equal(dx[a:e], ops.conv3x3_bs_bwd_data(dy[a:e].contiguous(), w, relu_src=xs)), (a, e)
            assert torch.equal(dxu[a:e], ops.conv3x3_bs_bwd_data(dyp[a:e].contiguous(), w, relu_src=None, idx=code[a:e].contiguous())), (a, e)


def test_bs_refuses_shapes_outside_its_domain():
    from clsurvey_amd import ops, _lib
    with pytest.raises(_lib.ClhipError):           # 3 input channels: not whole 32-channel pairs
        ops.conv3x3_bs_fwd(torch.zeros(1, 3, 8, 8, device="cuda"), torch.zeros(64, 3, 3, 3, device="cuda"), torch.zeros(64, device="cuda"))
    with pytest.raises(_lib.ClhipError):           # 32 output channels: not a whole 64-channel group
        ops.conv3x3_bs_fwd(torch.zeros(1, 32, 8, 8, device="cuda"), torch.zeros(32, 32, 3, 3, device="cuda"), torch.zeros(32, device="cuda"))
    with pytest.raises(_lib.ClhipError):           # fused pooling on an odd map
        ops.conv3x3_bs_fwd(torch.zeros(1, 32, 13, 13, device="cuda"), torch.zeros(64, 32, 3, 3, device="cuda"), torch.zeros(64, device="cuda"),
                           pool=True)


@pytest.mark.parametrize("C,K", [(64, 64), (128, 256), (512, 512)])
def test_bs_error_is_that_of_an_fp32_chain(C, K):
    """Against an fp64 convolution on the same fp32 inputs, in units of sum|x w| per output (what an fp32 rounding analysis bounds):
    the split path's error must not exceed the direct f32 MFMA kernel's nor the Winograd kernel's on the same data (the claim
    `dtype: f32` rests on this), under 4 units of fp32's 2^-24 = 6e-8 in the maximum over ~10^5 outputs, under half a unit rms."""
    from clsurvey_amd import ops
    N, H, W = 4, 16, 16
    gen = np.random.RandomState(C + K)
    x = torch.from_numpy(np.maximum(gen.standard_normal((N, C, H, W)), 0).astype(np.float32))          # a ReLU output
    w = torch.from_numpy((gen.standard_normal((K, C, 3, 3)) * (2.0 / (9 * C)) ** 0.5).astype(np.float32))
    b = torch.zeros(K)
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    scale = F.conv2d(x.double().abs(), w.double().abs(), None, padding=1).clamp_min(1e-30)
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()

    def err(y):
        e = (y.double().cpu() - ref).abs() / scale
        return float(e.max()), float((e * e).mean().sqrt())
    bs, direct, wino = err(ops.conv3x3_bs_fwd(xd, wd, bd, relu=False)), err(ops.conv3x3_fwd(xd, wd, bd, False)), err(ops.conv3x3_wino_fwd(xd, wd, bd, relu=False))
    print("C=%d K=%d  error / sum|x w| (max, rms): bf16-split %.3e %.3e   direct f32 MFMA %.3e %.3e   Winograd f32 %.3e %.3e"
          % ((C, K) + bs + direct + wino))
    # (the leading product a0 b0 and the five small products are summed in accumulators of their own: measured a third of an fp32
    # chain's error, profiles/r05_bf16_split_dot.txt)
    assert bs[0] <= 2.5e-7 and bs[1] <= 2.5e-8
    assert bs[1] <= direct[1] and bs[0] <= 1.25 * direct[0]
    assert bs[1] <= wino[1]
    # backward-data likewise
    dy = torch.from_numpy(gen.standard_normal((N, K, H, W)).astype(np.float32))
    refd = F.conv_transpose2d(dy.double(), w.double(), padding=1)
    scaled = F.conv_transpose2d(dy.double().abs(), w.double().abs(), padding=1).clamp_min(1e-30)
    ed = ((ops.conv3x3_bs_bwd_data(dy.cuda(), wd).double().cpu() - refd).abs() / scaled)
    assert float(ed.max()) <= 2.5e-7 and float((ed * ed).mean().sqrt()) <= 2.5e-8


def test_bs_weight_image_is_an_exact_split():
    """The weight image holds three bf16 pieces per weight, lane-ordered as the B operand of v_mfma_f32_32x32x16_bf16:
    [n tile][16-channel chunk][tap][piece][lane = 32 * (k half) + (out channel & 31)][8 channels].  The pieces must sum to the
    fp32 weight EXACTLY (in fp64), piece i must be the bf16 rounding of what pieces < i left, and backward-data's image holds
    w[ci][ko] rotated by 180 degrees."""
    from clsurvey_amd import _lib, ops
    L = _lib.lib()
    dev = torch.device("cuda:0")
    C, K = 64, 128
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(K, C, 3, 3, generator=g) * 0.1).to(dev)
    x = torch.randn(1, C, 8, 8, generator=g).to(dev)
    for mode in (0, 1):
        ws = torch.zeros(L.clhip_conv3x3_bs_ws(C, K), dtype=torch.uint8, device=dev)
        if mode == 0:
            y = torch.empty(1, K, 8, 8, device=dev)
            assert L.clhip_conv3x3_bs_fwd(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), None, 1, C, K, 8, 8, 0, ws.data_ptr(), ws.numel(), None) == 0
            Ko, Ci = K, C
        else:
            dy = torch.randn(1, K, 8, 8, device=dev)
            dx = torch.empty(1, C, 8, 8, device=dev)
            assert L.clhip_conv3x3_bs_bwd_data(dy.data_ptr(), None, w.data_ptr(), None, dx.data_ptr(), 1, C, K, 8, 8, ws.data_ptr(), ws.numel(), None) == 0
            Ko, Ci = C, K
        torch.cuda.synchronize()
        n_nt, n_ch = Ko // 32, Ci // 16
        img = ws[: n_nt * n_ch * 27 * 1024].view(torch.int16).view(n_nt, n_ch, 9, 3, 2, 32, 8).cpu()      # [nt][chunk][tap][piece][kh][ko & 31][e]
        pieces = (img.to(torch.int32) << 16).view(torch.float32).double()                                   # bf16 bits -> value
        # -> [piece][ko][ci][tap]
        val = pieces.permute(3, 0, 5, 1, 4, 6, 2).reshape(3, Ko, Ci, 9)
        wc = w.cpu().double()
        want = wc.reshape(K, C, 9) if mode == 0 else wc.reshape(K, C, 9).flip(2).permute(1, 0, 2)
        assert torch.equal(val.sum(0), want)
        assert torch.equal(val[0].float(), want.float().to(torch.bfloat16).float())
        assert torch.equal(val[1].float(), (want - val[0]).float().to(torch.bfloat16).float())


@pytest.mark.parametrize("mode", ["0", "2"])
def test_engine_parity_with_bs_off_and_everywhere(mode):
    """CLHIP_BS (read once per process by the plan builder): 0 = no layer takes the bf16-split kernels, 2 = every layer that can
    (default 1: the large maps only).  The engine's own parity tests — golden G1 and the full-size small_VGG9 step against the
    oracle — must hold on both sides of the switch."""
    env = dict(os.environ, CLHIP_BS=mode)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q",
                        "-p", "no:cacheprovider", "-k", "golden_g1 or (full_size_vs_oracle and small) or fused_conv_relu_pool"],
                       env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("mode", ["1", "2"])
def test_engine_parity_with_weight_gradients_on_a_side_stream(mode):
    """CLHIP_WGRAD_OVERLAP (read at plan creation): 1 = the weight-gradient launches of every conv layer on a side stream, 2 = of
    the small-map layers only (default 0: one stream; measured slower).  Same kernels, same order per stream: the engine's
    parity tests must hold, and the weight gradients stay bitwise deterministic."""
    env = dict(os.environ, CLHIP_WGRAD_OVERLAP=mode)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q",
                        "-p", "no:cacheprovider", "-k", "golden_g1 or (full_size_vs_oracle and small) or deterministic"],
                       env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("shape", [(2, 64, 192, 27, 27), (3, 32, 64, 12, 20), (2, 64, 64, 16, 16), (1, 192, 64, 27, 27), (128, 64, 192, 27, 27)])
def test_bs_5x5_forward_and_backward_data(shape):
    """The same kernel with 5 x 5 taps (torchvision AlexNet's features[3], models/net.py:96-125: Conv2d(64, 192, 5, padding=2) + ReLU)
    against torch CPU, at AlexNet's own size against the f32 LDS-halo kernel behind clhip_conv2d_fwd / _bwd_data."""
    from clsurvey_amd import ops
    N, C, K, H, W = shape
    gen = np.random.RandomState(N * 5 + C + K + H)
    x = torch.from_numpy(gen.standard_normal((N, C, H, W)).astype(np.float32))
    w = torch.from_numpy((gen.standard_normal((K, C, 5, 5)) * (2.0 / (25 * C)) ** 0.5).astype(np.float32))
    b = torch.from_numpy((gen.standard_normal((K,)) * 0.1).astype(np.float32))
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    big = N >= 100
    z_ref = ops.conv2d_fwd(xd, wd, bd, 1, 2, False) if big else F.conv2d(x, w, b, padding=2)
    assert _rel(ops.conv5x5_bs_fwd(xd, wd, bd, relu=False), z_ref) <= 2e-5
    assert _rel(ops.conv5x5_bs_fwd(xd, wd, bd, relu=True), z_ref.clamp_min(0)) <= 2e-5
    if C % 64 or K % 32:
        return
    dy = torch.from_numpy(gen.standard_normal((N, K, H, W)).astype(np.float32))
    msrc = torch.from_numpy(gen.standard_normal((N, C, H, W)).astype(np.float32))
    dyd, md = dy.cuda(), msrc.cuda()
    dx_ref = ops.conv2d_bwd_data(dyd, wd, (N, C, H, W), 1, 2) if big else F.conv_transpose2d(dy, w, padding=2)
    assert _rel(ops.conv5x5_bs_bwd_data(dyd, wd), dx_ref) <= 2e-5
    assert _rel(ops.conv5x5_bs_bwd_data(dyd, wd, relu_src=md), dx_ref.cuda() * (md > 0)) <= 2e-5
