"""torch-CPU references for the convolution parity tests: the reference's own arithmetic (nn.Conv2d / MaxPool2d of models/VGGSlim.py:27-40
and their autograd backward) at EVERY batch size, the bench batch of 200 included — a 200 x 512 x 8 x 8 layer is ~1 s on the host."""
import torch
import torch.nn.functional as F


def conv(x, w, b):
    return F.conv2d(x, w, b, padding=w.shape[-1] // 2)


def bwd_data(dy, w):
    return F.conv_transpose2d(dy, w, padding=w.shape[-1] // 2)


def unpool(dyp, code):
    """Gradient of a 2x2 max-pool from the pooled gradient and the arg-max codes (0..3 = position 2 * row + column inside the window,
    4 = dead window: no positive maximum after ReLU, csrc/common.hpp), on the CPU."""
    dyp, code = dyp.cpu(), code.cpu()
    N, K, h, w = dyp.shape
    out = torch.zeros((N, K, 2 * h, 2 * w), dtype=dyp.dtype)
    for c in range(4):
        out[:, :, (c >> 1)::2, (c & 1)::2] = torch.where(code == c, dyp, torch.zeros_like(dyp))
    return out


def bwd_weight(x, dy, ksize=3):
    """(dW, db) of a stride-1 'same' convolution by autograd on the CPU."""
    K, C = dy.shape[1], x.shape[1]
    w = torch.zeros(K, C, ksize, ksize, dtype=x.dtype, requires_grad=True)
    b = torch.zeros(K, dtype=x.dtype, requires_grad=True)
    F.conv2d(x.cpu(), w, b, padding=ksize // 2).backward(dy.cpu())
    return w.grad, b.grad
