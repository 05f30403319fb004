import numpy as np
import torch


class ClassErrorMeter:
    def __init__(self, topk=(1,), accuracy=False):
        self.topk = sorted(topk)
        self.accuracy = accuracy
        self.reset()

    def reset(self):
        self.sum = {k: 0 for k in self.topk}
        self.n = 0

    def add(self, output, target):
        if torch.is_tensor(output):
            output = output.detach().cpu().numpy()
        if torch.is_tensor(target):
            target = target.detach().cpu().numpy()
        if output.ndim == 1:
            output = output[None]
        maxk = self.topk[-1]
        pred = np.argsort(-output, axis=1, kind="stable")[:, :maxk]
        correct = pred == target.reshape(-1, 1)
        for k in self.topk:
            self.sum[k] += output.shape[0] - correct[:, :k].sum()
        self.n += output.shape[0]

    def value(self, k=-1):
        if k != -1:
            err = float(self.sum[k]) / self.n * 100.0
            return 100.0 - err if self.accuracy else err
        return [self.value(k_) for k_ in self.topk]
