"""Oracle: VGGSlim forward / backward (torch-CPU fp32 restatement).

Restates /root/reference/src/models/VGGSlim.py:27-40 (make_layers: 3x3 pad-1
conv + ReLU, 2x2 stride-2 max-pool), :43-76 (VGGSlim: features -> Identity
avgpool -> flatten (NCHW order) -> Linear-ReLU-Linear-ReLU-Linear) and the
losses used on the path:
  * CrossEntropyLoss (mean)                      EWC/train_EWC.py:183
  * nll_loss(log_softmax, size_average=False)    EWC/main_EWC.py:148
  * MSELoss(size_average=False) vs zeros         MAS/train_MAS.py:556-560
"""
import torch
import torch.nn.functional as F

CFGS = {  # models/VGGSlim.py:19-23
    "small_VGG9": [64, "M", 64, "M", 64, 64, "M", 128, 128, "M"],
    "base_VGG9": [64, "M", 64, "M", 128, 128, "M", 256, 256, "M"],
    "wide_VGG9": [64, "M", 128, "M", 256, 256, "M", 512, 512, "M"],
    "deep_VGG22": [64, "M", 64, 64, 64, 64, 64, 64, "M", 128, 128, 128, 128, 128, 128, "M",
                   256, 256, 256, 256, 256, 256, "M"],
}


def n_conv(cfg):
    return sum(1 for v in cfg if v != "M")


def init_params(cfg, fc_dims, num_classes, in_hw, gen, in_ch=3):
    """Deterministic init with the statistics of torchvision VGG._initialize_weights
    (kaiming_normal fan_out for conv, N(0, .01) linear, zero bias) driven by a
    numpy RandomState so it reproduces on any box."""
    import numpy as np
    params = []
    c = in_ch
    hw = in_hw
    for v in cfg:
        if v == "M":
            hw //= 2
            continue
        std = (2.0 / (v * 9)) ** 0.5
        params.append(torch.from_numpy((gen.standard_normal((v, c, 3, 3)) * std).astype(np.float32)))
        params.append(torch.from_numpy((gen.standard_normal((v,)) * 0.01).astype(np.float32)))
        c = v
    d = c * hw * hw
    for o in list(fc_dims) + [num_classes]:
        params.append(torch.from_numpy((gen.standard_normal((o, d)) * 0.01).astype(np.float32)))
        params.append(torch.from_numpy((gen.standard_normal((o,)) * 0.01).astype(np.float32)))
        d = o
    return params


def features(params, cfg, x):
    """The feature extractor alone (conv / ReLU / max-pool stack), un-flattened."""
    i = 0
    for v in cfg:
        if v == "M":
            x = F.max_pool2d(x, kernel_size=2, stride=2)
        else:
            x = F.relu(F.conv2d(x, params[i], params[i + 1], padding=1))
            i += 2
    return x


def forward(params, cfg, x, gates=None):
    """params: flat list [conv_w, conv_b]*, [fc_w, fc_b]*3 in module order."""
    i = 0
    for v in cfg:
        if v == "M":
            x = F.max_pool2d(x, kernel_size=2, stride=2)
        else:
            x = F.relu(F.conv2d(x, params[i], params[i + 1], padding=1))
            i += 2
    x = torch.flatten(x, 1)
    nfc = (len(params) - i) // 2
    for j in range(nfc):
        x = F.linear(x, params[i], params[i + 1])
        if j < nfc - 1:
            x = F.relu(x)
        i += 2
    return x


def loss_fn(logits, y, kind):
    if kind == "ce_mean":
        return F.cross_entropy(logits, y)
    if kind == "ce_sum":
        return F.nll_loss(F.log_softmax(logits, dim=1), y, reduction="sum")
    if kind == "mse_sum_zero":
        return (logits ** 2).sum()
    raise ValueError(kind)


def loss_and_grads(params, cfg, x, y, kind="ce_mean", need_dx=False):
    ps = [p.detach().clone().requires_grad_(True) for p in params]
    xx = x.detach().clone().requires_grad_(need_dx)
    logits = forward(ps, cfg, xx)
    loss = loss_fn(logits, y, kind)
    outs = torch.autograd.grad(loss, ps + ([xx] if need_dx else []))
    grads = list(outs[:len(ps)])
    dx = outs[len(ps)] if need_dx else None
    return logits.detach(), loss.detach(), grads, dx
