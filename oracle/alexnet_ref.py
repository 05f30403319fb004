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
