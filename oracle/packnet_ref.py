"""Oracle: PackNet masks and masked SGD (numpy; uint8 masks are compared bit-exactly).

Restates methods/packnet/prune.py (SparsePruner) and methods/packnet/packnetSGD.py.
Mask value = owning task index (1-based), 0 = free / pruned.
"""
import numpy as np


def make_finetuning_mask(mask, cur):
    """prune.py:141-155: free weights (0) are handed to the current task."""
    m = mask.copy()
    m[m == 0] = cur
    return m


def cutoff_rank(prune_perc, numel):
    """prune.py:32: Python round() (banker's rounding) of perc * numel."""
    return round(prune_perc * numel)


def pruning_mask(weights, mask, cur, prune_perc):
    """prune.py:24-52: k-th smallest |w| among the current task's weights (kthvalue, 1-based k);
    every current-task weight with |w| <= cutoff is released (ties all go)."""
    sel = np.abs(weights[mask == cur]).astype(np.float32).ravel()
    k = cutoff_rank(prune_perc, sel.size)
    if k < 1:
        raise ValueError("kthvalue: k out of range (reference would raise too)")
    cutoff = np.partition(sel, k - 1)[k - 1]
    m = mask.copy()
    m[(np.abs(weights) <= cutoff) & (mask == cur)] = 0
    return m, np.float32(cutoff), k


def prune(weights, mask, cur, prune_perc):
    """prune.py:54-71: new mask + pruned weights set to 0."""
    m, cutoff, k = pruning_mask(weights, mask, cur, prune_perc)
    w = weights.copy()
    w[m == 0] = 0.0
    return w, m, cutoff, k


def make_grads_zero(grad, mask, cur):
    """prune.py:73-97 (weights): grads of weights not owned by the current task are zeroed."""
    g = grad.copy()
    g[mask != cur] = 0
    return g


def make_pruned_zero(weights, mask):
    """prune.py:99-106."""
    w = weights.copy()
    w[mask == 0] = 0.0
    return w


def apply_mask(weights, mask, dataset_idx):
    """prune.py:108-118: keep only weights of tasks 1..dataset_idx."""
    w = weights.copy()
    w[mask == 0] = 0.0
    w[mask > dataset_idx] = 0.0
    return w


def packnet_sgd_step(theta, grad, buf, lr, momentum, wd, first):
    """packnetSGD.py:35-56: weight decay only where grad != 0; momentum SGD."""
    theta = theta.astype(np.float32)
    d = grad.astype(np.float32).copy()
    if wd != 0:
        d = d + (np.float32(wd) * theta) * (grad != 0).astype(np.float32)
    if momentum != 0:
        buf = d.copy() if first or buf is None else (buf * np.float32(momentum) + d).astype(np.float32)
        d = buf
    return (theta - np.float32(lr) * d).astype(np.float32), buf
