"""CPU restatement of EBLL's two objectives — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows src/methods/EBLL:
  AlexNet_EBLL.py:9-26       AutoEncoder: code = sigmoid(W_e f + b_e), reconstruction = W_d code + b_d
  Finetune_SGD_EBLL.py:139-160  stage 1: total = alpha * MSELoss(reconstruction, f) + CrossEntropy(classifier(reconstruction))
  Finetune_SGD_EBLL.py:497      optim.Adadelta(autoencoder.parameters(), lr)   (rho 0.9, eps 1e-6)
  Finetune_SGD_EBLL.py:297-317  stage 2: total = lambda * sum distillation(old heads) + CrossEntropy(new head)
                                         + reg_alpha * sum MSELoss(code_i, code_i of the frozen previous model)
Features / classifier come in as plain functions so that this file does not depend on a model class.
Pinned by tests/golden/G16_ebll.npz (generated from the reference by tests/golden/make_g16.py).
"""
import torch
import torch.nn.functional as F

from .lwf_ref import distillation_loss


def encode(f, w_e, b_e):
    return torch.sigmoid(F.linear(f, w_e, b_e))


def autoencode(f, w_e, b_e, w_d, b_d):
    return F.linear(encode(f, w_e, b_e), w_d, b_d)


def stage1_objective(f, labels, ae, classifier_tail, alpha):
    """ae = (w_e, b_e, w_d, b_d); classifier_tail(features) -> logits of the last head.  Returns (task, encoder) losses."""
    recon = autoencode(f, *ae)
    return F.cross_entropy(classifier_tail(recon), labels), F.mse_loss(recon, f), recon


def adadelta_step(params, grads, state, lr, rho=0.9, eps=1e-6):
    """In place; state = [(square_avg, acc_delta)] per parameter (zeros at the start)."""
    with torch.no_grad():
        for p, g, (sq, acc) in zip(params, grads, state):
            sq.mul_(rho).addcmul_(g, g, value=1 - rho)
            delta = (acc + eps).sqrt() / (sq + eps).sqrt() * g
            acc.mul_(rho).addcmul_(delta, delta, value=1 - rho)
            p.sub_(lr * delta)


def stage2_objective(head_outputs, codes, labels, target_logits, target_codes, T, lam, reg_alpha):
    """Returns (task CE, lambda * distillation, UNSCALED code loss); total = task + dist + reg_alpha * code."""
    task = F.cross_entropy(head_outputs[-1], labels)
    dist = sum(distillation_loss(o, t, T) for o, t in zip(head_outputs[:-1], target_logits))
    code = sum(F.mse_loss(c, t) for c, t in zip(codes, target_codes))
    return task, lam * dist, code
