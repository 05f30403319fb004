"""Oracle: HAT forward / criterion / back-masks / HAT_SGD step (torch-CPU fp32 restatement).

Restates methods/HAT/networks/vgg_hat.py:83-127 (forward, mask; alexnet_hat.py:4-13 is the same net with the Dropout in
front of each gated Linear layer), :258-295 (get_view_for),
approaches/hat.py:58-89 (init_masks), :285-299 (criterion) and HAT_utils.py:192-250 (HAT_SGD.step).
Parameters come as dicts keyed like the reference's named_parameters():
  convs.<i>.weight/bias, conv_embs.<i>.weight, fcs.<i>.weight/bias, fc_embs.<i>.weight,
  classifier.0.weight/bias
"""
import torch
import torch.nn.functional as F


def n_layers(P):
    nc = len([k for k in P if k.startswith("convs.") and k.endswith("weight")])
    nf = len([k for k in P if k.startswith("fcs.") and k.endswith("weight")])
    return nc, nf


def masks(P, t, s):
    """vgg_hat.py:121-127: gate = sigmoid(s * E[t]) per conv / fc layer."""
    nc, nf = n_layers(P)
    out = [torch.sigmoid(s * P["conv_embs.%d.weight" % i][t]) for i in range(nc)]
    out += [torch.sigmoid(s * P["fc_embs.%d.weight" % i][t]) for i in range(nf)]
    return out


def forward(P, pool_after, t, x, s, conv_geo=None, pool=(2, 2), drop=None, first_drop=False, decisions=None, pre=None):
    """vgg_hat.py:83-119. pool_after: set of conv indices followed by the net's max-pool.
    conv_geo: (stride, padding) per convolution (default: the VGG 3x3 / 1 / 1); pool: (kernel, stride) of the one
    MaxPool2d module the net shares (vgg_hat.py:41-44); drop: one keep-mask (already scaled by 1 / (1 - p)) per gated
    Linear layer, or None in eval mode; first_drop: alexnet_hat.py:12-13 — relu(fc(drop(x))) — instead of the VGG
    order drop(relu(fc(x))) (vgg_hat.py:110-114).
    decisions (optional, test infrastructure): one dict per conv / gated Linear layer, {'mask': bool like the layer's
    output after its pool, 'idx': int64 window position r*k + c for pooled layers} — the ReLU / arg-max DECISIONS of
    another evaluation of the same step; the layer is then evaluated on that piecewise-linear branch (relu -> pool as
    gather(z, idx) * mask, see oracle.alexnet_ref.forward_forced).  pre (optional list): receives every pre-activation."""
    nc, nf = n_layers(P)
    mk = masks(P, t, s)
    for i in range(nc):
        stride, pad = conv_geo[i] if conv_geo is not None else (1, 1)
        z = F.conv2d(x, P["convs.%d.weight" % i], P["convs.%d.bias" % i], stride=stride, padding=pad)
        if pre is not None:
            pre.append(z)
        if decisions is None:
            x = F.relu(z)
            if i in pool_after:
                x = F.max_pool2d(x, pool[0], pool[1])
        else:
            d = decisions[i]
            if i in pool_after:
                win = z.unfold(2, pool[0], pool[1]).unfold(3, pool[0], pool[1])
                win = win.reshape(win.shape[0], win.shape[1], win.shape[2], win.shape[3], pool[0] * pool[0])
                z = torch.gather(win, 4, d["idx"].unsqueeze(-1)).squeeze(-1)
            x = z * d["mask"].to(z.dtype)
        x = x * mk[i].view(1, -1, 1, 1)
    x = x.reshape(x.shape[0], -1)
    for i in range(nf):
        if drop is not None and first_drop:
            x = x * drop[i]
        z = F.linear(x, P["fcs.%d.weight" % i], P["fcs.%d.bias" % i])
        if pre is not None:
            pre.append(z)
        x = F.relu(z) if decisions is None else z * decisions[nc + i]["mask"].to(z.dtype)
        if drop is not None and not first_drop:
            x = x * drop[i]
        x = x * mk[nc + i]
    return F.linear(x, P["classifier.0.weight"], P["classifier.0.bias"]), mk


def criterion(logits, y, mk, mask_pre, lamb):
    """hat.py:285-299."""
    reg, count = 0.0, 0.0
    if mask_pre is not None:
        for m, mp in zip(mk, mask_pre):
            aux = 1 - mp
            reg = reg + (m * aux).sum()
            count = count + aux.sum()
    else:
        for m in mk:
            reg = reg + m.sum()
            count = count + m.numel()
    reg = reg / count
    return F.cross_entropy(logits, y) + lamb * reg, lamb * reg


def init_masks(P, current_task, smax):
    """hat.py:58-89: a^{<t} = max over previous tasks of sigmoid(smax*E[t']); back-masks 1 - a^{<t}."""
    mask_pre = None
    for t in range(current_task):
        m = [v.detach().clone() for v in masks(P, t, smax)]
        mask_pre = m if mask_pre is None else [torch.max(a, b) for a, b in zip(mask_pre, m)]
    mask_back = {}
    if mask_pre is not None:
        for n in P:
            v = get_view_for(P, n, mask_pre)
            if v is not None:
                mask_back[n] = 1 - v
    return mask_pre, mask_back


def get_view_for(P, n, mk):
    """vgg_hat.py:258-295."""
    nc, nf = n_layers(P)
    conv_m, fc_m = mk[:nc], mk[nc:]
    segs = n.split(".")
    if len(segs) != 3 or segs[0] not in ("convs", "fcs"):
        return None
    idx = int(segs[1])
    w = P[n]
    if segs[0] == "convs":
        if n == "convs.0.weight":
            return conv_m[0].view(-1, 1, 1, 1).expand_as(w)
        if segs[2] == "weight":
            return torch.min(conv_m[idx].view(-1, 1, 1, 1).expand_as(w), conv_m[idx - 1].view(1, -1, 1, 1).expand_as(w))
        return conv_m[idx].view(-1)
    if n == "fcs.0.weight":
        smid2 = w.shape[1] // conv_m[-1].numel()
        pre = conv_m[-1].view(-1, 1).expand(-1, smid2).contiguous().view(1, -1).expand_as(w)
        return torch.min(fc_m[0].view(-1, 1).expand_as(w), pre)
    if segs[2] == "weight":
        return torch.min(fc_m[idx].view(-1, 1).expand_as(w), fc_m[idx - 1].view(1, -1).expand_as(w))
    return fc_m[idx].view(-1)


def hat_sgd_step(name, theta, grad, buf, mask_back, t, s, smax, lr, momentum, wd, thres_cosh=50.0, clipgrad=10000.0,
                 finetune=False, first=True):
    """HAT_utils.py:211-248 for one parameter."""
    g = grad.clone()
    if wd != 0 and "embs" not in name:
        g = g + wd * theta
    if t > 0 and name in mask_back:
        g = g * mask_back[name]
    if not finetune:
        if "embs" in name:
            num = torch.cosh(torch.clamp(s * theta, -thres_cosh, thres_cosh)) + 1
            den = torch.cosh(theta) + 1
            g = g * (smax / s * num / den)
        norm = float(g.norm(2))
        coef = clipgrad / (norm + 1e-6)
        if coef < 1:
            g = g * coef
    if momentum != 0:
        buf = g.clone() if first or buf is None else buf * momentum + g
        d = buf
    else:
        d = g
    return theta - lr * d, buf, g
