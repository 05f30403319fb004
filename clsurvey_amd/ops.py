"""Thin torch-facing wrappers over the C ABI (include/clhip.h).

Every function takes contiguous fp32 CUDA(HIP) tensors, passes raw device pointers and the
current HIP stream to libclhip, and returns torch tensors.  There is no CPU path: a CPU tensor
raises.  The autograd.Function classes at the bottom make the kernels usable from ordinary
nn.Module code (HAT / GEM wrappers); the hot loops use clsurvey_amd.net.NetEngine instead.
"""
import torch

from . import _lib
from ._lib import check

_ws_cache = {}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("clsurvey_amd ops need HIP device tensors (no CPU fallback)")
        if not t.is_contiguous():
            raise RuntimeError("clsurvey_amd ops need contiguous tensors")


def _ptr(t):
    return None if t is None else t.data_ptr()


def workspace(nbytes, device, tag="default"):
    """Grow-only scratch buffer per (device, tag)."""
    key = (str(device), tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


# ------------------------------------------------------------------ convolution
def conv3x3_fwd(x, w, b, relu=True):
    _chk(x, w, b)
    N, C, H, W = x.shape
    K = w.shape[0]
    assert w.shape == (K, C, 3, 3) and x.dtype == torch.float32
    y = torch.empty((N, K, H, W), dtype=torch.float32, device=x.device)
    check(_lib.lib().clhip_conv3x3_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(y), N, C, K, H, W, int(relu), _stream()),
          "clhip_conv3x3_fwd")
    return y


def conv3x3_wino_fwd(x, w, b, relu=True, pool=False):
    """clhip_conv3x3_wino_fwd: the forward operators above through Winograd F(2x2, 3x3).  Returns y, or (y_pool, idx_u8)."""
    _chk(x, w, b)
    N, C, H, W = x.shape
    K = w.shape[0]
    L = _lib.lib()
    ws = torch.empty(L.clhip_conv3x3_wino_ws(C, K), dtype=torch.uint8, device=x.device)
    if pool:
        y = torch.empty((N, K, H // 2, W // 2), dtype=torch.float32, device=x.device)
        idx = torch.empty((N, K, H // 2, W // 2), dtype=torch.uint8, device=x.device)
    else:
        y, idx = torch.empty((N, K, H, W), dtype=torch.float32, device=x.device), None
    check(L.clhip_conv3x3_wino_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(y), _ptr(idx) if pool else None, N, C, K, H, W, int(relu),
                                   _ptr(ws), ws.numel(), _stream()), "clhip_conv3x3_wino_fwd")
    return (y, idx) if pool else y


def conv3x3_wino_bwd_data(dy, w, relu_src=None, idx=None):
    """clhip_conv3x3_wino_bwd_data: dx of the 3x3 convolution (idx: dy is the POOLED gradient + arg-max codes)."""
    _chk(dy, w)
    K, C = w.shape[0], w.shape[1]
    N = dy.shape[0]
    H, W = (dy.shape[2] * 2, dy.shape[3] * 2) if idx is not None else (dy.shape[2], dy.shape[3])
    L = _lib.lib()
    ws = torch.empty(L.clhip_conv3x3_wino_ws(C, K), dtype=torch.uint8, device=dy.device)
    dx = torch.empty((N, C, H, W), dtype=torch.float32, device=dy.device)
    check(L.clhip_conv3x3_wino_bwd_data(_ptr(dy), _ptr(idx) if idx is not None else None, _ptr(w),
                                        _ptr(relu_src) if relu_src is not None else None, _ptr(dx), N, C, K, H, W, _ptr(ws),
                                        ws.numel(), _stream()), "clhip_conv3x3_wino_bwd_data")
    return dx


def conv3x3_bs_fwd(x, w, b, relu=True, pool=False):
    """clhip_conv3x3_bs_fwd: the forward operators above on the bf16 matrix cores with split fp32 operands (csrc/bsconv.hip).
    Returns y, or (y_pool, idx_u8)."""
    _chk(x, w, b)
    N, C, H, W = x.shape
    K = w.shape[0]
    L = _lib.lib()
    ws = torch.empty(L.clhip_conv3x3_bs_ws(C, K), dtype=torch.uint8, device=x.device)
    if pool:
        y = torch.empty((N, K, H // 2, W // 2), dtype=torch.float32, device=x.device)
        idx = torch.empty((N, K, H // 2, W // 2), dtype=torch.uint8, device=x.device)
    else:
        y, idx = torch.empty((N, K, H, W), dtype=torch.float32, device=x.device), None
    check(L.clhip_conv3x3_bs_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(y), _ptr(idx) if pool else None, N, C, K, H, W, int(relu),
                                 _ptr(ws), ws.numel(), _stream()), "clhip_conv3x3_bs_fwd")
    return (y, idx) if pool else y


def conv3x3_bs_bwd_data(dy, w, relu_src=None, idx=None):
    """clhip_conv3x3_bs_bwd_data: dx of the 3x3 convolution on the same path (idx: dy is the POOLED gradient + arg-max codes)."""
    _chk(dy, w)
    K, C = w.shape[0], w.shape[1]
    N = dy.shape[0]
    H, W = (dy.shape[2] * 2, dy.shape[3] * 2) if idx is not None else (dy.shape[2], dy.shape[3])
    L = _lib.lib()
    ws = torch.empty(L.clhip_conv3x3_bs_ws(C, K), dtype=torch.uint8, device=dy.device)
    dx = torch.empty((N, C, H, W), dtype=torch.float32, device=dy.device)
    check(L.clhip_conv3x3_bs_bwd_data(_ptr(dy), _ptr(idx) if idx is not None else None, _ptr(w),
                                      _ptr(relu_src) if relu_src is not None else None, _ptr(dx), N, C, K, H, W, _ptr(ws),
                                      ws.numel(), _stream()), "clhip_conv3x3_bs_bwd_data")
    return dx


def conv5x5_bs_fwd(x, w, b, relu=True):
    """clhip_conv5x5_bs_fwd: 5 x 5 / stride 1 / padding 2 convolution (+ bias, ReLU) on the bf16-split kernel (csrc/bsconv.hip)."""
    _chk(x, w, b)
    N, C, H, W = x.shape
    K = w.shape[0]
    assert w.shape == (K, C, 5, 5)
    L = _lib.lib()
    ws = torch.empty(L.clhip_conv5x5_bs_ws(C, K), dtype=torch.uint8, device=x.device)
    y = torch.empty((N, K, H, W), dtype=torch.float32, device=x.device)
    check(L.clhip_conv5x5_bs_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(y), N, C, K, H, W, int(relu), _ptr(ws), ws.numel(), _stream()),
          "clhip_conv5x5_bs_fwd")
    return y


def conv5x5_bs_bwd_data(dy, w, relu_src=None):
    """clhip_conv5x5_bs_bwd_data: dx of that convolution (x (relu_src > 0) when given)."""
    _chk(dy, w)
    K, C = w.shape[0], w.shape[1]
    N, _, H, W = dy.shape
    L = _lib.lib()
    ws = torch.empty(L.clhip_conv5x5_bs_ws(C, K), dtype=torch.uint8, device=dy.device)
    dx = torch.empty((N, C, H, W), dtype=torch.float32, device=dy.device)
    check(L.clhip_conv5x5_bs_bwd_data(_ptr(dy), _ptr(w), _ptr(relu_src) if relu_src is not None else None, _ptr(dx), N, C, K, H, W,
                                      _ptr(ws), ws.numel(), _stream()), "clhip_conv5x5_bs_bwd_data")
    return dx


def conv3x3_wino_bwd_weight(x, dy, idx=None):
    """clhip_conv3x3_wino_bwd_weight: (dw, db) of the 3x3 convolution (idx: dy is the POOLED gradient + arg-max codes)."""
    _chk(x, dy)
    N, C, H, W = x.shape
    K = dy.shape[1]
    L = _lib.lib()
    ws = torch.empty(L.clhip_conv3x3_wino_bwd_weight_ws(N, C, K, H, W), dtype=torch.uint8, device=x.device)
    dw = torch.empty((K, C, 3, 3), dtype=torch.float32, device=x.device)
    db = torch.empty((K,), dtype=torch.float32, device=x.device)
    check(L.clhip_conv3x3_wino_bwd_weight(_ptr(x), _ptr(dy), _ptr(idx) if idx is not None else None, _ptr(dw), _ptr(db), N, C, K, H, W,
                                          _ptr(ws), ws.numel(), _stream()), "clhip_conv3x3_wino_bwd_weight")
    return dw, db


def conv3x3_bs_bwd_weight(x, dy, idx=None):
    """clhip_conv3x3_bs_bwd_weight: (dw, db) on the bf16 matrix cores with split fp32 operands (idx: dy is the POOLED gradient + codes)."""
    _chk(x, dy, idx)
    N, C, H, W = x.shape
    K = dy.shape[1]
    L = _lib.lib()
    ws = torch.empty(max(L.clhip_conv3x3_bs_bwd_weight_ws(N, C, K, H, W), 16), dtype=torch.uint8, device=x.device)
    dw = torch.empty((K, C, 3, 3), dtype=torch.float32, device=x.device)
    db = torch.empty((K,), dtype=torch.float32, device=x.device)
    check(L.clhip_conv3x3_bs_bwd_weight(_ptr(x), _ptr(dy), _ptr(idx) if idx is not None else None, _ptr(dw), _ptr(db), N, C, K, H, W,
                                        _ptr(ws), ws.numel(), _stream()), "clhip_conv3x3_bs_bwd_weight")
    return dw, db


def conv5x5_bs_bwd_weight(x, dy):
    """clhip_conv5x5_bs_bwd_weight: (dw, db) of the 5x5 / padding-2 convolution on the bf16 matrix cores with split fp32 operands."""
    _chk(x, dy)
    N, C, H, W = x.shape
    K = dy.shape[1]
    L = _lib.lib()
    ws = torch.empty(max(L.clhip_conv5x5_bs_bwd_weight_ws(N, C, K, H, W), 16), dtype=torch.uint8, device=x.device)
    dw = torch.empty((K, C, 5, 5), dtype=torch.float32, device=x.device)
    db = torch.empty((K,), dtype=torch.float32, device=x.device)
    check(L.clhip_conv5x5_bs_bwd_weight(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), N, C, K, H, W, _ptr(ws), ws.numel(), _stream()),
          "clhip_conv5x5_bs_bwd_weight")
    return dw, db


def conv3x3_wino_bwd(x, dy, w, relu_src=None, idx=None):
    """clhip_conv3x3_wino_bwd: (dx, dw, db) of one 3x3 layer as ONE grid (backward-data and weight-gradient blocks interleaved).
    Returns None for a layer the merged grid does not take (CLHIP_ENOTSUP: nothing was launched)."""
    _chk(x, dy, w)
    N, C, H, W = x.shape
    K = dy.shape[1]
    L = _lib.lib()
    ws = torch.empty(max(L.clhip_conv3x3_wino_bwd_ws(N, C, K, H, W), 16), dtype=torch.uint8, device=x.device)
    dx = torch.empty((N, C, H, W), dtype=torch.float32, device=x.device)
    dw = torch.empty((K, C, 3, 3), dtype=torch.float32, device=x.device)
    db = torch.empty((K,), dtype=torch.float32, device=x.device)
    rc = L.clhip_conv3x3_wino_bwd(_ptr(x), _ptr(dy), _ptr(idx) if idx is not None else None, _ptr(w),
                                  _ptr(relu_src) if relu_src is not None else None, _ptr(dx), _ptr(dw), _ptr(db), N, C, K, H, W,
                                  _ptr(ws), ws.numel(), _stream())
    if rc == -3:                      # CLHIP_ENOTSUP
        return None
    check(rc, "clhip_conv3x3_wino_bwd")
    return dx, dw, db


def conv3x3_relu_pool_fwd(x, w, b):
    """fused conv + bias + ReLU + 2x2 max-pool: returns (y_pool, idx_u8).  idx codes are 0..4, NOT an index to gather with:
    0..3 = window position r * 2 + c of the first maximum (ATen's order), 4 (CLHIP_POOL_DEAD, csrc/common.hpp) = the window's
    maximum after ReLU is not positive — no gradient passes through it.  Only the fused-unpool consumers
    (conv3x3_bwd_weight_unpool / conv3x3_bwd_data_unpool, which test `code == position`) take these codes;
    maxpool_bwd (general k x k) takes the plain arg-max bytes of maxpool_fwd."""
    _chk(x, w, b)
    N, C, H, W = x.shape
    K = w.shape[0]
    y = torch.empty((N, K, H // 2, W // 2), dtype=torch.float32, device=x.device)
    idx = torch.empty((N, K, H // 2, W // 2), dtype=torch.uint8, device=x.device)
    check(_lib.lib().clhip_conv3x3_relu_pool_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(y), _ptr(idx), N, C, K, H, W, _stream()),
          "clhip_conv3x3_relu_pool_fwd")
    return y, idx


def conv3x3_bwd_weight_unpool(x, dy_pool, idx, need_bias=True):
    _chk(x, dy_pool, idx)
    N, C, H, W = x.shape
    K = dy_pool.shape[1]
    L = _lib.lib()
    ws = workspace(L.clhip_conv3x3_bwd_weight_ws(N, C, K, H, W), x.device, "wgrad")
    dw = torch.empty((K, C, 3, 3), dtype=torch.float32, device=x.device)
    db = torch.empty((K,), dtype=torch.float32, device=x.device) if need_bias else None
    check(L.clhip_conv3x3_bwd_weight_unpool(_ptr(x), _ptr(dy_pool), _ptr(idx), _ptr(dw), _ptr(db), N, C, K, H, W,
                                            _ptr(ws), ws.numel(), _stream()), "clhip_conv3x3_bwd_weight_unpool")
    return dw, db


def conv3x3_bwd_data(dy, w, relu_src=None):
    _chk(dy, w, relu_src)
    N, K, H, W = dy.shape
    C = w.shape[1]
    dx = torch.empty((N, C, H, W), dtype=torch.float32, device=dy.device)
    check(_lib.lib().clhip_conv3x3_bwd_data(_ptr(dy), _ptr(w), _ptr(relu_src), _ptr(dx), N, C, K, H, W, _stream()),
          "clhip_conv3x3_bwd_data")
    return dx


def conv3x3_bwd_data_unpool(dy_pool, idx, w, relu_src=None):
    """backward-data from the POOLED gradient + arg-max codes (fused max-pool backward); raises ClhipError
    (CLHIP_ENOTSUP) on shapes outside the 16-byte staging path."""
    _chk(dy_pool, idx, w, relu_src)
    N, K, Hp, Wp = dy_pool.shape
    C = w.shape[1]
    dx = torch.empty((N, C, 2 * Hp, 2 * Wp), dtype=torch.float32, device=dy_pool.device)
    check(_lib.lib().clhip_conv3x3_bwd_data_unpool(_ptr(dy_pool), _ptr(idx), _ptr(w), _ptr(relu_src), _ptr(dx), N, C, K,
                                                   2 * Hp, 2 * Wp, _stream()), "clhip_conv3x3_bwd_data_unpool")
    return dx


def conv3x3_bwd_weight(x, dy, need_bias=True):
    _chk(x, dy)
    N, C, H, W = x.shape
    K = dy.shape[1]
    L = _lib.lib()
    nbytes = L.clhip_conv3x3_bwd_weight_ws(N, C, K, H, W)
    ws = workspace(nbytes, x.device, "wgrad")
    dw = torch.empty((K, C, 3, 3), dtype=torch.float32, device=x.device)
    db = torch.empty((K,), dtype=torch.float32, device=x.device) if need_bias else None
    check(L.clhip_conv3x3_bwd_weight(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), N, C, K, H, W, _ptr(ws), ws.numel(),
                                     _stream()), "clhip_conv3x3_bwd_weight")
    return dw, db


def conv3x3_bwd_weight_slabs(x, dy, idx=None):
    """First half of conv3x3_bwd_weight as the plan executor issues it: per-split partial sums into the 'wgrad' workspace.
    Returns (workspace, splits) for conv3x3_bwd_weight_reduce."""
    import ctypes
    _chk(x, dy, idx)
    N, C, H, W = x.shape
    K = dy.shape[1]
    L = _lib.lib()
    ws = workspace(L.clhip_conv3x3_bwd_weight_ws(N, C, K, H, W), x.device, "wgrad")
    splits = ctypes.c_int(0)
    check(L.clhip_conv3x3_bwd_weight_slabs(_ptr(x), _ptr(dy), _ptr(idx), N, C, K, H, W, _ptr(ws), ws.numel(),
                                           ctypes.byref(splits), _stream()), "clhip_conv3x3_bwd_weight_slabs")
    return ws, splits.value


def conv3x3_bwd_weight_reduce(ws, splits, K, C, need_bias=True):
    dw = torch.empty((K, C, 3, 3), dtype=torch.float32, device=ws.device)
    db = torch.empty((K,), dtype=torch.float32, device=ws.device) if need_bias else None
    check(_lib.lib().clhip_conv3x3_bwd_weight_reduce(_ptr(ws), _ptr(dw), _ptr(db), K, C, int(splits), _stream()),
          "clhip_conv3x3_bwd_weight_reduce")
    return dw, db


def _out_hw(H, W, R, S, stride, pad):
    return (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1


def conv2d_fwd(x, w, b, stride=1, pad=0, relu=False):
    """General convolution (AlexNet layers); the 3x3 pad-1 stride-1 case belongs to conv3x3_fwd."""
    _chk(x, w, b)
    N, C, H, W = x.shape
    K, _, R, S = w.shape
    OH, OW = _out_hw(H, W, R, S, stride, pad)
    y = torch.empty((N, K, OH, OW), dtype=torch.float32, device=x.device)
    check(_lib.lib().clhip_conv2d_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(y), N, C, H, W, K, R, S, int(stride), int(pad), int(relu),
                                      _stream()), "clhip_conv2d_fwd")
    return y


def conv2d_bwd_data(dy, w, x_shape, stride=1, pad=0, relu_src=None):
    _chk(dy, w, relu_src)
    N, C, H, W = x_shape
    K, _, R, S = w.shape
    dx = torch.empty((N, C, H, W), dtype=torch.float32, device=dy.device)
    check(_lib.lib().clhip_conv2d_bwd_data(_ptr(dy), _ptr(w), _ptr(relu_src), _ptr(dx), N, C, H, W, K, R, S, int(stride), int(pad),
                                           _stream()), "clhip_conv2d_bwd_data")
    return dx


def conv2d_bwd_weight(x, dy, ksize, stride=1, pad=0, need_bias=True):
    _chk(x, dy)
    N, C, H, W = x.shape
    K = dy.shape[1]
    R, S = ksize
    L = _lib.lib()
    ws = workspace(L.clhip_conv2d_bwd_weight_ws(N, C, H, W, K, R, S, int(stride), int(pad)), x.device, "wgrad2d")
    dw = torch.empty((K, C, R, S), dtype=torch.float32, device=x.device)
    db = torch.empty((K,), dtype=torch.float32, device=x.device) if need_bias else None
    check(L.clhip_conv2d_bwd_weight(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), N, C, H, W, K, R, S, int(stride), int(pad), _ptr(ws),
                                    ws.numel(), _stream()), "clhip_conv2d_bwd_weight")
    return dw, db


def conv2d_s2d_fwd(x, w, b, stride, pad, relu=False, ws=None):
    """Strided convolution through space-to-depth + the dense 3x3 kernels (csrc/s2dconv.hip); returns (y, ws) — hand `ws` to
    conv2d_s2d_bwd_weight(None, ...) to reuse the phase planes.  None when the shape is not taken."""
    _chk(x, w, b)
    N, C, H, W = x.shape
    K, _, R, S = w.shape
    L = _lib.lib()
    nbytes = L.clhip_conv2d_s2d_ws(N, C, H, W, K, R, int(stride), int(pad)) if R == S else 0
    if not nbytes:
        return None
    if ws is None:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    OH, OW = _out_hw(H, W, R, S, stride, pad)
    y = torch.empty((N, K, OH, OW), dtype=torch.float32, device=x.device)
    check(L.clhip_conv2d_s2d_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(y), N, C, H, W, K, R, int(stride), int(pad), int(relu), _ptr(ws),
                                 ws.numel(), _stream()), "clhip_conv2d_s2d_fwd")
    return y, ws


def conv2d_s2d_bwd_weight(x, dy, x_shape, ksize, stride, pad, ws=None):
    """(dW, db) of the same layer; x None = the phase planes of conv2d_s2d_fwd are still in `ws`."""
    _chk(dy, x)
    N, C, H, W = x_shape
    K = dy.shape[1]
    R = int(ksize)
    L = _lib.lib()
    nbytes = L.clhip_conv2d_s2d_ws(N, C, H, W, K, R, int(stride), int(pad))
    if not nbytes:
        return None
    if ws is None:
        assert x is not None
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dy.device)
    dw = torch.empty((K, C, R, R), dtype=torch.float32, device=dy.device)
    db = torch.empty((K,), dtype=torch.float32, device=dy.device)
    check(L.clhip_conv2d_s2d_bwd_weight(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), N, C, H, W, K, R, int(stride), int(pad), _ptr(ws),
                                        ws.numel(), _stream()), "clhip_conv2d_s2d_bwd_weight")
    return dw, db


# ------------------------------------------------------------------ batch norm
def bn_fwd(z, gamma, beta, running_mean, running_var, training, momentum, eps, relu):
    """nn.BatchNorm2d (+ReLU): returns (y, save_mean, save_invstd); training mode moves the running statistics in place."""
    _chk(z, gamma, beta, running_mean, running_var)
    N, C, H, W = z.shape
    y = torch.empty_like(z)
    mean = torch.empty(C, dtype=torch.float32, device=z.device)
    invstd = torch.empty(C, dtype=torch.float32, device=z.device)
    L = _lib.lib()
    ws = workspace(L.clhip_bn_ws(C), z.device, "bn")
    check(L.clhip_bn_fwd(_ptr(z), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), _ptr(y), _ptr(mean), _ptr(invstd),
                         N, C, H * W, int(training), float(momentum), float(eps), int(relu), _ptr(ws), ws.numel(), _stream()),
          "clhip_bn_fwd")
    return y, mean, invstd


def bn_bwd(dy, y, z, gamma, mean, invstd, training, relu):
    """(dz, dgamma, dbeta) of bn_fwd."""
    _chk(dy, y, z, gamma, mean, invstd)
    N, C, H, W = z.shape
    dz = torch.empty_like(z)
    dgamma = torch.empty(C, dtype=torch.float32, device=z.device)
    dbeta = torch.empty(C, dtype=torch.float32, device=z.device)
    L = _lib.lib()
    ws = workspace(L.clhip_bn_ws(C), z.device, "bn")
    check(L.clhip_bn_bwd(_ptr(dy), _ptr(y), _ptr(z), _ptr(gamma), _ptr(mean), _ptr(invstd), _ptr(dz), _ptr(dgamma), _ptr(dbeta),
                         N, C, H * W, int(training), int(relu), _ptr(ws), ws.numel(), _stream()), "clhip_bn_bwd")
    return dz, dgamma, dbeta


# ------------------------------------------------------------------ pooling
def maxpool_fwd(x, k, stride):
    _chk(x)
    N, C, H, W = x.shape
    OH, OW = (H - k) // stride + 1, (W - k) // stride + 1
    y = torch.empty((N, C, OH, OW), dtype=torch.float32, device=x.device)
    idx = torch.empty((N, C, OH, OW), dtype=torch.uint8, device=x.device)
    check(_lib.lib().clhip_maxpool_fwd(_ptr(x), _ptr(y), _ptr(idx), N * C, H, W, int(k), int(stride), _stream()), "clhip_maxpool_fwd")
    return y, idx


def maxpool_bwd(dy, idx, x_shape, k, stride):
    _chk(dy, idx)
    N, C, H, W = x_shape
    dx = torch.empty((N, C, H, W), dtype=torch.float32, device=dy.device)
    check(_lib.lib().clhip_maxpool_bwd(_ptr(dy), _ptr(idx), _ptr(dx), N * C, H, W, int(k), int(stride), _stream()), "clhip_maxpool_bwd")
    return dx


def maxpool2_fwd(x):
    _chk(x)
    N, C, H, W = x.shape
    y = torch.empty((N, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
    idx = torch.empty((N, C, H // 2, W // 2), dtype=torch.uint8, device=x.device)
    check(_lib.lib().clhip_maxpool2_fwd(_ptr(x), _ptr(y), _ptr(idx), N * C, H, W, _stream()), "clhip_maxpool2_fwd")
    return y, idx


def maxpool2_bwd(dy, idx):
    _chk(dy, idx)
    N, C, OH, OW = dy.shape
    dx = torch.empty((N, C, OH * 2, OW * 2), dtype=torch.float32, device=dy.device)
    check(_lib.lib().clhip_maxpool2_bwd(_ptr(dy), _ptr(idx), _ptr(dx), N * C, OH * 2, OW * 2, _stream()),
          "clhip_maxpool2_bwd")
    return dx


# ------------------------------------------------------------------ fully connected
def fc_fwd(x, w, b, relu=False):
    _chk(x, w, b)
    M, I = x.shape
    O = w.shape[0]
    L = _lib.lib()
    ws = workspace(L.clhip_fc_ws(M, I, O), x.device, "fc")
    y = torch.empty((M, O), dtype=torch.float32, device=x.device)
    check(L.clhip_fc_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(y), M, I, O, int(relu), _ptr(ws), ws.numel(), _stream()),
          "clhip_fc_fwd")
    return y


def fc_bwd_data(dy, w, relu_src=None):
    _chk(dy, w, relu_src)
    M, O = dy.shape
    I = w.shape[1]
    L = _lib.lib()
    ws = workspace(L.clhip_fc_ws(M, I, O), dy.device, "fc")
    dx = torch.empty((M, I), dtype=torch.float32, device=dy.device)
    check(L.clhip_fc_bwd_data(_ptr(dy), _ptr(w), _ptr(relu_src), _ptr(dx), M, I, O, _ptr(ws), ws.numel(), _stream()),
          "clhip_fc_bwd_data")
    return dx


def fc_bwd_weight(x, dy, need_bias=True):
    _chk(x, dy)
    M, I = x.shape
    O = dy.shape[1]
    L = _lib.lib()
    ws = workspace(L.clhip_fc_ws(M, I, O), x.device, "fc")
    dw = torch.empty((O, I), dtype=torch.float32, device=x.device)
    db = torch.empty((O,), dtype=torch.float32, device=x.device) if need_bias else None
    check(L.clhip_fc_bwd_weight(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), M, I, O, _ptr(ws), ws.numel(), _stream()),
          "clhip_fc_bwd_weight")
    return dw, db


def relu_bwd(dy, y):
    _chk(dy, y)
    dx = torch.empty_like(dy)
    check(_lib.lib().clhip_relu_bwd(_ptr(dy), _ptr(y), _ptr(dx), dy.numel(), _stream()), "clhip_relu_bwd")
    return dx


# ------------------------------------------------------------------ losses
def softmax_ce(logits, labels, reduction="mean", stats=None):
    """returns (loss[1], dlogits). stats: optional float64[2] device tensor accumulating
    (sum of batch losses, #correct)."""
    _chk(logits, labels, stats)
    N, Cc = logits.shape
    assert labels.dtype == torch.int64
    dl = torch.empty_like(logits)
    loss = torch.empty((1,), dtype=torch.float32, device=logits.device)
    red = {"mean": 0, "sum": 1}[reduction]
    check(_lib.lib().clhip_softmax_ce(_ptr(logits), _ptr(labels), N, Cc, red, _ptr(dl), _ptr(loss), _ptr(stats),
                                      _stream()), "clhip_softmax_ce")
    return loss, dl


def mse_zero_sum(logits):
    _chk(logits)
    dl = torch.empty_like(logits)
    loss = torch.empty((1,), dtype=torch.float32, device=logits.device)
    check(_lib.lib().clhip_mse_zero_sum(_ptr(logits), logits.numel(), _ptr(dl), _ptr(loss), _stream()),
          "clhip_mse_zero_sum")
    return loss, dl


# ------------------------------------------------------------------ optimizers / importance (in place)
def reg_sgd_step(theta, grad, omega, init_val, buf, reg_lambda, lr, momentum, wd, first):
    _chk(theta, grad, omega, init_val, buf)
    check(_lib.lib().clhip_reg_sgd_step(_ptr(theta), _ptr(grad), _ptr(omega), _ptr(init_val), _ptr(buf), theta.numel(),
                                        float(reg_lambda), float(lr), float(momentum), float(wd), int(first), _stream()),
          "clhip_reg_sgd_step")


def fisher_accum(omega, grad, data_len):
    _chk(omega, grad)
    check(_lib.lib().clhip_fisher_accum(_ptr(omega), _ptr(grad), omega.numel(), float(data_len), _stream()),
          "clhip_fisher_accum")


def mas_accum(omega, grad, batch_index, batch_size):
    _chk(omega, grad)
    prev = float(batch_index * batch_size)
    curr = float((batch_index + 1) * batch_size)
    check(_lib.lib().clhip_mas_accum(_ptr(omega), _ptr(grad), omega.numel(), prev, curr, _stream()), "clhip_mas_accum")


def si_step(theta, grad, omega, init_val, w, buf, reg_lambda, lr, momentum, wd, first):
    _chk(theta, grad, omega, init_val, w, buf)
    check(_lib.lib().clhip_si_step(_ptr(theta), _ptr(grad), _ptr(omega), _ptr(init_val), _ptr(w), _ptr(buf),
                                   theta.numel(), float(reg_lambda), float(lr), float(momentum), float(wd), int(first),
                                   _stream()), "clhip_si_step")


def si_consolidate(omega, w, theta, init_val, slack=1e-3):
    _chk(omega, w, theta, init_val)
    check(_lib.lib().clhip_si_consolidate(_ptr(omega), _ptr(w), _ptr(theta), _ptr(init_val), omega.numel(), float(slack),
                                          _stream()), "clhip_si_consolidate")


# ------------------------------------------------------------------ autograd bridges
class Conv3x3ReLUFn(torch.autograd.Function):
    """relu(conv3x3(x, w) + b) — nn.Conv2d + nn.ReLU(inplace) pair of VGGSlim.py:34-38."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        x = x.contiguous()
        y = conv3x3_fwd(x, w.contiguous(), b.contiguous() if b is not None else None, relu)
        ctx.relu = relu
        ctx.save_for_backward(x, w, y)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.relu:
            dy = relu_bwd(dy, y)
        dx = conv3x3_bwd_data(dy, w) if ctx.needs_input_grad[0] else None
        dw, db = conv3x3_bwd_weight(x, dy, ctx.has_bias)
        return dx, dw, db, None


class Conv2dReLUFn(torch.autograd.Function):
    """relu?(conv2d(x, w, stride, pad) + b) for the non-3x3 layers (torchvision alexnet features)."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, relu):
        x = x.contiguous()
        y = conv2d_fwd(x, w.contiguous(), b.contiguous() if b is not None else None, stride, pad, relu)
        ctx.cfg = (stride, pad, relu, b is not None)
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        stride, pad, relu, has_bias = ctx.cfg
        dy = dy.contiguous()
        if relu:
            dy = relu_bwd(dy, y)
        dx = conv2d_bwd_data(dy, w, tuple(x.shape), stride, pad) if ctx.needs_input_grad[0] else None
        dw, db = conv2d_bwd_weight(x, dy, tuple(w.shape[2:]), stride, pad, has_bias)
        return dx, dw, db, None, None, None


class BatchNormReLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, training, momentum, eps, relu):
        z = z.contiguous()
        y, mean, invstd = bn_fwd(z, gamma.contiguous(), beta.contiguous(), running_mean, running_var, training, momentum, eps, relu)
        ctx.cfg = (bool(training), bool(relu))
        ctx.save_for_backward(z, y, gamma, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, y, gamma, mean, invstd = ctx.saved_tensors
        training, relu = ctx.cfg
        dz, dgamma, dbeta = bn_bwd(dy.contiguous(), y, z, gamma.contiguous(), mean, invstd, training, relu)
        return dz, dgamma, dbeta, None, None, None, None, None, None


def sigmoid_fwd(x):
    _chk(x)
    y = torch.empty_like(x)
    check(_lib.lib().clhip_sigmoid_fwd(_ptr(x), _ptr(y), x.numel(), _stream()), "clhip_sigmoid_fwd")
    return y


def sigmoid_bwd(dy, y):
    _chk(dy, y)
    dx = torch.empty_like(y)
    check(_lib.lib().clhip_sigmoid_bwd(_ptr(dy), _ptr(y), _ptr(dx), y.numel(), _stream()), "clhip_sigmoid_bwd")
    return dx


def mse_mean(a, b, grad_scale=1.0, want_grad=True):
    """nn.MSELoss()(a, b) -> (loss[1] device tensor, grad_scale * d loss / d a or None)."""
    _chk(a, b)
    assert a.shape == b.shape
    loss = torch.empty(1, dtype=torch.float32, device=a.device)
    da = torch.empty_like(a) if want_grad else None
    check(_lib.lib().clhip_mse_mean(_ptr(a), _ptr(b), a.numel(), float(grad_scale), _ptr(da), _ptr(loss), _stream()), "clhip_mse_mean")
    return loss, da


class SigmoidFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = sigmoid_fwd(x.contiguous())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return sigmoid_bwd(dy.contiguous(), y)


class MseMeanFn(torch.autograd.Function):
    """nn.MSELoss()(a, target): gradient w.r.t. a only (the target is data)."""

    @staticmethod
    def forward(ctx, a, target):
        loss, da = mse_mean(a.contiguous(), target.contiguous())
        ctx.save_for_backward(da)
        return loss.view(())

    @staticmethod
    def backward(ctx, dl):
        (da,) = ctx.saved_tensors
        return da * dl, None


class MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, stride):
        x = x.contiguous()
        y, idx = maxpool_fwd(x, k, stride)
        ctx.cfg = (tuple(x.shape), k, stride)
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        shape, k, stride = ctx.cfg
        return maxpool_bwd(dy.contiguous(), idx, shape, k, stride), None, None


class MaxPool2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y, idx = maxpool2_fwd(x.contiguous())
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        return maxpool2_bwd(dy.contiguous(), idx)


class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, relu):
        x = x.contiguous()
        y = fc_fwd(x, w.contiguous(), b.contiguous() if b is not None else None, relu)
        ctx.relu = relu
        ctx.has_bias = b is not None
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.relu:
            dy = relu_bwd(dy, y)
        dx = fc_bwd_data(dy, w) if ctx.needs_input_grad[0] else None
        dw, db = fc_bwd_weight(x, dy, ctx.has_bias)
        return dx, dw, db, None


class SoftmaxCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, reduction):
        loss, dl = softmax_ce(logits.contiguous(), labels, reduction)
        ctx.save_for_backward(dl)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None


def conv3x3_relu(x, w, b, relu=True):
    return Conv3x3ReLUFn.apply(x, w, b, relu)


def maxpool2(x):
    return MaxPool2Fn.apply(x)


def conv2d_relu(x, w, b, stride=1, pad=0, relu=True):
    return Conv2dReLUFn.apply(x, w, b, stride, pad, relu)


def batchnorm_relu(z, bn, relu=True):
    """nn.BatchNorm2d module `bn` (+ReLU) on the HIP kernels; follows bn.training like the module itself."""
    if bn.training:
        bn.num_batches_tracked += 1
    return BatchNormReLUFn.apply(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training, bn.momentum, bn.eps, relu)


def sigmoid(x):
    return SigmoidFn.apply(x)


def mse_loss(a, target):
    return MseMeanFn.apply(a, target)


def maxpool(x, k, stride):
    return MaxPoolFn.apply(x, k, stride)


def linear(x, w, b, relu=False):
    return LinearFn.apply(x, w, b, relu)


def cross_entropy(logits, labels, reduction="mean"):
    return SoftmaxCEFn.apply(logits, labels, reduction)
