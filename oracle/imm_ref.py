"""CPU restatement of IMM's merge and precision estimate — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows src/methods/IMM/merge.py:
  merge_mean  :222-241  as executed: returns the task's own tensor (reference bug, see merge_mean); intended: the average
  merge_mode  :226-228  sum_m (F_m / S) * theta_m, S = sum of the precisions of tasks 0..idx
  diag_fisher :155-183  precision = 1e-8 + sum over phases and batches of grad(mean nll(targets))^2 / #batches(phase)
                        (targets are SAMPLED from the softmax in the reference; here they are an argument)
Pinned by tests/golden/G13_imm.npz (generated from the reference by tests/golden/make_g13.py).
"""
import torch

from . import vgg_ref


def merge_mean(thetas):
    """What merge.py:205-241 DOES in mean mode: inside the loop it rebinds the loop variable `param_value` to a
    state_dict tensor of a preceding model (:223-224), so the final `param_value.data = mean_param.clone()` (:239) lands
    on that temporary and the merged model keeps the deep copy of the task's own model: mean-IMM == the unmerged task
    model (pinned by G13)."""
    return thetas[-1].clone()


def merge_mean_intended(thetas):
    """The formula the code was meant to apply (running sum from zero, divided by the number of models)."""
    acc = torch.zeros_like(thetas[0])
    for t in thetas:
        acc = acc + t
    return acc / len(thetas)


def merge_mode(thetas, precisions, sum_precision):
    acc = torch.zeros_like(thetas[0])
    for t, f in zip(thetas, precisions):
        acc += (f / sum_precision) * t
    return acc


def diag_fisher(params, cfg, phases, targets, exclude=()):
    """phases: list of lists of x batches; targets: same structure (int64). Returns per-parameter precision list
    (None for excluded indices)."""
    prec = [None if i in exclude else torch.zeros_like(p) + 1e-8 for i, p in enumerate(params)]
    for xs, ys in zip(phases, targets):
        for x, y in zip(xs, ys):
            _, _, grads, _ = vgg_ref.loss_and_grads(params, cfg, x, y, "ce_mean")
            for i, g in enumerate(grads):
                if prec[i] is not None:
                    prec[i] += g ** 2 / len(xs)
    return prec
