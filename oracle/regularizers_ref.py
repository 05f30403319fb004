"""Oracle: importance-weight passes and penalised optimizers (torch-CPU fp32).

All functions operate on flat lists of tensors (one per parameter) and mutate
nothing; they return new tensors.  Order of floating-point operations follows
the reference line by line so results agree to the last bit on CPU.
"""
import torch

from . import vgg_ref


# --------------------------------------------------------------------- SGD core
def _momentum_update(d, buf, momentum, first):
    # optim.SGD semantics used by every reference optimizer, dampening 0,
    # nesterov False: EWC/train_EWC.py:70-81
    if momentum == 0:
        return d, buf
    if first or buf is None:
        buf = d.clone()
    else:
        buf = buf * momentum + d
    return buf, buf


def reg_sgd_step(theta, grad, omega, init_val, buf, reg_lambda, lr, momentum, wd, first):
    """Weight_Regularized_SGD.step — EWC/train_EWC.py:23-86 (== MAS/train_MAS.py:32-95).
    omega/init_val None  <=>  `p not in reg_params` (new head: plain SGD)."""
    d = grad.clone()
    if omega is not None:
        weight_dif = theta - init_val                      # :62
        regulizer = weight_dif * (2 * reg_lambda * omega)  # :64
        d = d + regulizer                                  # :65
    if wd != 0:
        d = d + wd * theta                                 # :70-71
    d, buf = _momentum_update(d, buf, momentum, first)
    theta = theta - lr * d                                 # :83
    return theta, buf


def fisher_accum(omega, grad, data_len):
    """diag_fisher inner update — EWC/main_EWC.py:155: omega += grad**2 / data_len"""
    return omega + grad ** 2 / data_len


def diag_fisher(params, cfg, batches, data_len):
    """EWC/main_EWC.py:138-157. batches: iterable of (x, y); loss is the SUM over the
    batch (size_average=False, :148) so grad is the batch-summed gradient, squared."""
    omega = [torch.zeros_like(p) for p in params]
    for x, y in batches:
        _, _, grads, _ = vgg_ref.loss_and_grads(params, cfg, x, y, "ce_sum")
        omega = [fisher_accum(o, g, data_len) for o, g in zip(omega, grads)]
    return omega


def mas_accum(omega, grad, batch_index, batch_size):
    """Objective_After_SGD.step — MAS/train_MAS.py:167-173."""
    prev_size = batch_index * batch_size
    curr_size = (batch_index + 1) * batch_size
    o = omega * prev_size
    o = o + grad.abs()
    return o / curr_size


def mas_importance(params, cfg, batches):
    """compute_importance_l2 — MAS/train_MAS.py:508-567: loss = sum(out**2) (:556-560),
    batch_size argument is labels.size(0) of the *current* batch (:563)."""
    omega = [torch.zeros_like(p) for p in params]
    for idx, (x, y) in enumerate(batches):
        _, _, grads, _ = vgg_ref.loss_and_grads(params, cfg, x, y, "mse_sum_zero")
        omega = [mas_accum(o, g, idx, y.shape[0]) for o, g in zip(omega, grads)]
    return omega


def si_step(theta, grad, omega, init_val, w, buf, reg_lambda, lr, momentum, wd, first):
    """Elastic_SGD.step — SI/train_SI.py:28-126."""
    unreg = grad.clone()                                    # :55
    theta0 = theta.clone()                                  # :63
    weight_dif = theta0 - init_val                          # :69
    d = grad + weight_dif * (2 * reg_lambda * omega)        # :71-73
    if wd != 0:
        d = d + wd * theta                                  # :82-83
    d, buf = _momentum_update(d, buf, momentum, first)      # :85-96
    theta = theta - lr * d                                  # :98
    w_diff = theta - theta0                                 # :99
    change = (w_diff * unreg) * -1                          # :103-105
    w = w + change                                          # :120
    return theta, buf, w


def si_consolidate(omega, w, theta, init_val, slack=1e-3):
    """update_reg_params — SI/train_SI.py:301-351 (the later identical redefinition
    :367-430 is the one bound at import)."""
    path_diff = theta - init_val
    dominator = path_diff.pow(2) + slack
    this_omega = torch.clamp(w / dominator, min=0)          # max(., 0) :346
    return omega + this_omega, torch.zeros_like(w), theta.clone()


# ----------------------------------------------------------------- LR schedules
def set_lr_trace(val_improved, lr, variant="ewc"):
    """Restates set_lr of EWC/train_EWC.py:89-101 ('ewc': stop at count > 10) and
    SI/train_SI.py:129-141 ('si': stop at count >= 10).  `val_improved` is the
    per-epoch sequence of booleans 'val acc beat best'.  Returns list of
    (epoch, lr_used, continued)."""
    count = 0
    out = []
    for ep, imp in enumerate(val_improved):
        cont = not (count > 10 if variant == "ewc" else count >= 10)
        if count == 5:
            lr = lr * 0.1
        out.append((ep, lr, cont))
        if not cont:
            break
        count = 0 if imp else count + 1
    return out
