"""CPU restatement of LwF's objective — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows src/methods/LwF/main_LWF.py:
  distillation_loss :47-76   student logits y, teacher logits t, temperature T:
                             loss = 1/N sum_rows [ log sum_j exp((y_j - max y)/T) - sum_j p_j (y_j - max y)/T ],
                             p = softmax(t - max t)^(1/T), renormalised
  lwf_objective     :184-202 CrossEntropy(mean) on the last head + lambda * sum over the old heads of distillation_loss
Pinned by tests/golden/G14_lwf.npz (generated from the reference by tests/golden/make_g14.py).
"""
import torch
import torch.nn.functional as F


def distillation_loss(y, t, T):
    ys = (y - y.max(1, keepdim=True)[0]) / T
    p = F.softmax(t - t.max(1, keepdim=True)[0], dim=1).pow(1.0 / T)
    p = p / p.sum(1, keepdim=True)
    return (torch.log(torch.exp(ys).sum(1)) - (p * ys).sum(1)).sum() / y.shape[0]


def lwf_objective(head_outputs, labels, teacher_logits, T, lam):
    """head_outputs: list of [N][C_h] (last = new task); teacher_logits: list for the old heads."""
    task = F.cross_entropy(head_outputs[-1], labels)
    dist = sum(distillation_loss(o, t, T) for o, t in zip(head_outputs[:-1], teacher_logits))
    return task, lam * dist
