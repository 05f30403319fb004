"""TEST INFRASTRUCTURE ONLY — CPU restatement of the AlexNet training pass of the reference (never imported by the product).

The reference runs torchvision's AlexNet (src/models/net.py:96-125) through torch's own CPU / cuDNN operators and
autograd; this file states the same computation functionally on torch CPU fp32 so that the parity tests can inject the
dropout masks the HIP engine used:

* nn.Dropout in training mode (classifier[0], [3]): y = x * m, m ~ Bernoulli(1-p) / (1-p) per element per sample;
* GEM's masks (src/methods/rehearsal/GEM/gem.py:166-196): one m of a single sample's feature shape per Dropout module,
  drawn at the first forward after `reset_dropout_config()` (gem.py:206-209: once per observe) and shared by the whole
  batch and by the memory passes of that observe.

Pinned by construction: the operators ARE the reference's backend (torch.nn.functional on CPU).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def forward(model, x, masks=None):
    """Forward of a features/classifier module tree on CPU with explicit dropout masks.

    masks: {index of the nn.Dropout among all Dropout modules in forward order: tensor broadcastable to its input}
    (None / missing = identity, i.e. eval mode)."""
    masks = masks or {}
    di = 0
    for m in model.features.children():
        if isinstance(m, nn.Dropout):
            if di in masks:
                x = x * masks[di]
            di += 1
        elif isinstance(m, nn.Conv2d):
            x = F.conv2d(x, m.weight, m.bias, m.stride, m.padding)
        elif isinstance(m, nn.BatchNorm2d):      # '_BN' variants (models/VGGSlim.py:35-36): the module's own semantics
            x = F.batch_norm(x, m.running_mean, m.running_var, m.weight, m.bias, m.training, m.momentum, m.eps)
        elif isinstance(m, nn.ReLU):
            x = F.relu(x)
        elif isinstance(m, nn.MaxPool2d):
            x = F.max_pool2d(x, m.kernel_size, m.stride)
        else:
            raise NotImplementedError(type(m))
    x = torch.flatten(x, 1)
    for m in model.classifier.children():
        if isinstance(m, nn.Dropout):
            if di in masks:
                x = x * masks[di]
            di += 1
        elif isinstance(m, nn.Linear):
            x = F.linear(x, m.weight, m.bias)
        elif isinstance(m, nn.ReLU):
            x = F.relu(x)
        else:
            raise NotImplementedError(type(m))
    return x


def loss_and_grads(model, x, y, masks=None):
    """(loss, logits, [grad per parameter in model.parameters() order]) of mean cross-entropy."""
    params = list(model.parameters())
    for p in params:
        p.grad = None
    logits = forward(model, x, masks)
    loss = F.cross_entropy(logits, y)
    grads = torch.autograd.grad(loss, params)
    return loss.detach(), logits.detach(), [g.detach() for g in grads]


def gem_observe(model, bufs, x, t, y, masks, memories, cum_nc, lr, margin, momentum=0.9):
    """One gem.Net.observe (gem.py:206-287) on a Dropout net with the given mask rows (shared by the memory passes and
    the current batch, gem.py:180-191), SGD(momentum, no weight decay) update included.

    memories: [(past_task, [(xb, yb), ...])] in observed order; bufs: momentum buffers per parameter (None at first).
    Returns (loss, hits, n_violations, G [P][len(memories)+1] float32 with the current task last)."""
    import numpy as np
    from . import gem_ref
    params = list(model.parameters())

    def offsets(task):
        return (0 if task == 0 else cum_nc[task - 1]), cum_nc[task]

    cols = []
    for past, batches in memories:
        o1, o2 = offsets(past)
        acc = [torch.zeros_like(p) for p in params]
        for xb, yb in batches:                       # gradients accumulate over the batches (sum of batch means)
            loss = F.cross_entropy(forward(model, xb, masks)[:, o1:o2], yb)
            acc = [a + g for a, g in zip(acc, torch.autograd.grad(loss, params))]
        cols.append(np.concatenate([a.numpy().reshape(-1) for a in acc]))
    o1, o2 = offsets(t)
    out = forward(model, x, masks)[:, o1:o2]
    loss = F.cross_entropy(out, y)
    hits = int((out.argmax(1) == y).sum())
    grads = [g.detach() for g in torch.autograd.grad(loss, params)]
    gflat = np.concatenate([g.numpy().reshape(-1) for g in grads])
    viol = 0
    G = np.stack(cols + [gflat], axis=1).astype(np.float32) if cols else gflat[:, None].astype(np.float32)
    if cols:
        dotp, viol = gem_ref.violations(G, len(cols), list(range(len(cols))))
        if viol:
            gflat, _ = gem_ref.project2cone2(gflat, G[:, :len(cols)], margin)
            grads = [torch.from_numpy(a.copy()) for a in gem_ref.overwrite_grad(gflat, [tuple(p.shape) for p in params])]
    sgd_momentum_step(params, grads, bufs, lr, momentum)
    return loss.detach(), hits, viol, G


def sgd_momentum_step(params, grads, bufs, lr, momentum=0.9):
    """torch.optim.SGD(momentum, dampening 0, weight_decay 0): buf = g at the first step, then momentum * buf + g."""
    with torch.no_grad():
        for i, (p, g) in enumerate(zip(params, grads)):
            bufs[i] = g.clone() if bufs[i] is None else bufs[i] * momentum + g
            p.sub_(lr * bufs[i])


# ------------------------------------------------------------------------------------------------ decision-forced evaluation
def _pool_windows(z, k, s):
    """[N,C,H,W] -> [N,C,OH,OW,k*k], window position r*k + c (the scan order of a k x k window)."""
    w = z.unfold(2, k, s).unfold(3, k, s)
    return w.reshape(w.shape[0], w.shape[1], w.shape[2], w.shape[3], k * k)


def forward_forced(model, x, masks, decisions):
    """`forward` with every non-linear DECISION taken from `decisions` instead of from the data (the module-tree form of
    oracle.vgg_ref.forward_forced): one dict per Conv2d / hidden Linear in forward order, {'mask': bool, shaped like that
    block's output (after its max-pool, if any)} for the ReLU, plus {'idx': int64 [N,C,OH,OW]} (window position r*k + c
    passed on) where a MaxPool2d follows.  Max-pooling commutes with ReLU, so relu -> pool is evaluated as
    gather(z, idx) * mask.  With the data's own decisions this IS `forward`; with another evaluation's decisions it is the
    piecewise-linear branch that evaluation took, so two fp32 evaluation orders that disagree on a near-tie can be
    compared at rounding level.  Returns (logits, [pre-activation per block, before pool / ReLU])."""
    masks = masks or {}
    di, b, pre = 0, 0, []
    feats = list(model.features.children())
    i = 0
    while i < len(feats):
        m = feats[i]
        if isinstance(m, nn.Dropout):
            if di in masks:
                x = x * masks[di].to(x.dtype)
            di += 1
            i += 1
            continue
        assert isinstance(m, nn.Conv2d), type(m)
        z = F.conv2d(x, m.weight, m.bias, m.stride, m.padding)
        i += 1
        if i < len(feats) and isinstance(feats[i], nn.BatchNorm2d):
            bn = feats[i]
            z = F.batch_norm(z, bn.running_mean.clone(), bn.running_var.clone(), bn.weight, bn.bias, bn.training, bn.momentum, bn.eps)
            i += 1
        relu = i < len(feats) and isinstance(feats[i], nn.ReLU)
        i += 1 if relu else 0
        pre.append(z)
        d = decisions[b]
        b += 1
        if i < len(feats) and isinstance(feats[i], nn.MaxPool2d):
            mp = feats[i]
            k = mp.kernel_size if isinstance(mp.kernel_size, int) else mp.kernel_size[0]
            s = mp.stride if isinstance(mp.stride, int) else mp.stride[0]
            z = torch.gather(_pool_windows(z, k, s), 4, d["idx"].unsqueeze(-1)).squeeze(-1)
            i += 1
        x = z * d["mask"].to(z.dtype) if relu else z
    x = torch.flatten(x, 1)
    cls = list(model.classifier.children())
    i = 0
    while i < len(cls):
        m = cls[i]
        if isinstance(m, nn.Dropout):
            if di in masks:
                x = x * masks[di].to(x.dtype)
            di += 1
            i += 1
            continue
        assert isinstance(m, nn.Linear), type(m)
        x = F.linear(x, m.weight, m.bias)
        i += 1
        if i < len(cls) and isinstance(cls[i], nn.ReLU):
            pre.append(x)
            x = x * decisions[b]["mask"].to(x.dtype)
            b += 1
            i += 1
    return x, pre


def loss_and_grads_forced(model, x, y, masks, decisions):
    """(loss, logits, grads, pre-activations) of mean cross-entropy on the branch `decisions` selects."""
    params = list(model.parameters())
    logits, pre = forward_forced(model, x, masks, decisions)
    loss = F.cross_entropy(logits, y)
    grads = torch.autograd.grad(loss, params)
    return loss.detach(), logits.detach(), [g.detach() for g in grads], [z.detach() for z in pre]


def own_decisions(model, pre):
    """The decisions the pre-activations `pre` of forward_forced imply themselves (first maximum of a window wins)."""
    out, b = [], 0
    feats = list(model.features.children())
    for i, m in enumerate(feats):
        if not isinstance(m, nn.Conv2d):
            continue
        z = pre[b]
        b += 1
        j = i + 1
        while j < len(feats) and isinstance(feats[j], (nn.BatchNorm2d, nn.ReLU)):
            j += 1
        if j < len(feats) and isinstance(feats[j], nn.MaxPool2d):
            mp = feats[j]
            k = mp.kernel_size if isinstance(mp.kernel_size, int) else mp.kernel_size[0]
            s = mp.stride if isinstance(mp.stride, int) else mp.stride[0]
            win = _pool_windows(z, k, s)
            idx = win.argmax(4)
            out.append({"idx": idx, "mask": torch.gather(win, 4, idx.unsqueeze(-1)).squeeze(-1) > 0, "k": k, "s": s})
        else:
            out.append({"mask": z > 0})
    for z in pre[b:]:
        out.append({"mask": z > 0})
    return out
