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


# ------------------------------------------------------------------------------------------------ decision-forced evaluation
def _windows(z):
    """[N,K,H,W] -> [N,K,H/2,W/2,4], window position r*2 + c (ATen's scan order inside a 2x2 window)."""
    n, k, h, w = z.shape
    return z.reshape(n, k, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, k, h // 2, w // 2, 4)


def forward_forced(params, cfg, x, decisions):
    """The same network with every non-linear DECISION taken from `decisions` instead of from the data: per conv block
    {'mask': bool [N,K,H',W']} (ReLU on / off per output element) and, for a pooled block, {'idx': int64 [N,K,H/2,W/2]}
    (which of the four window elements is passed on); per hidden Linear {'mask': bool [N,D]}.  Where the decisions are the
    network's own this IS the network (ReLU(z) = z * [z > 0], max-pool = gather at the arg-max); with another
    evaluation's decisions it is the piecewise-linear branch that evaluation computed — so two fp32 evaluation orders
    that disagree on a near-tie can still be compared at rounding level.  Returns (logits, pre-activations)."""
    i, b, pre = 0, 0, []
    vs = list(cfg)
    j = 0
    while j < len(vs):
        z = F.conv2d(x, params[i], params[i + 1], padding=1)
        i += 2
        d = decisions[b]
        b += 1
        pooled = j + 1 < len(vs) and vs[j + 1] == "M"
        pre.append(z)
        if pooled:
            z = torch.gather(_windows(z), 4, d["idx"].unsqueeze(-1)).squeeze(-1)
            j += 1
        x = z * d["mask"].to(z.dtype)
        j += 1
    x = torch.flatten(x, 1)
    nfc = (len(params) - i) // 2
    for f in range(nfc):
        x = F.linear(x, params[i], params[i + 1])
        i += 2
        if f < nfc - 1:
            pre.append(x)
            x = x * decisions[b]["mask"].to(x.dtype)
            b += 1
    return x, pre


def loss_and_grads_forced(params, cfg, x, y, kind, decisions):
    ps = [p.detach().clone().requires_grad_(True) for p in params]
    logits, pre = forward_forced(ps, cfg, x, decisions)
    loss = loss_fn(logits, y, kind)
    grads = torch.autograd.grad(loss, ps)
    return logits.detach(), loss.detach(), list(grads), [z.detach() for z in pre]


def own_decisions(cfg, pre):
    """The decisions the data itself implies for the pre-activations `pre` of forward_forced (first maximum wins)."""
    out, b, j, vs = [], 0, 0, list(cfg)
    while j < len(vs):
        z = pre[b]
        if j + 1 < len(vs) and vs[j + 1] == "M":
            win = _windows(z)
            idx = win.argmax(4)               # torch returns the first maximal index
            out.append({"idx": idx, "mask": torch.gather(win, 4, idx.unsqueeze(-1)).squeeze(-1) > 0})
            j += 1
        else:
            out.append({"mask": z > 0})
        b += 1
        j += 1
    for z in pre[b:]:
        out.append({"mask": z > 0})
    return out
