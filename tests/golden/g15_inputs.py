"""Deterministic net / weights / batches shared by the G15 fixture generator (reference run, dev container) and the tests:
a small AlexNet-STRUCTURED module (torchvision.models.alexnet's layer types and geometry, narrow widths) for 3x67x67
inputs, numpy-driven weights and batches.  Plain torch modules only."""
import numpy as np
import torch
import torch.nn as nn

WIDTHS, FC, NCLS0, HW = (16, 24, 32, 32, 16), 64, 4, 67
N_OUT, NC_PER_TASK, N_MEM, BATCH = 8, [4, 4], 4, 6


class SmallAlexNet(nn.Module):
    def __init__(self, num_classes=NCLS0):
        super().__init__()
        c1, c2, c3, c4, c5 = WIDTHS
        self.features = nn.Sequential(
            nn.Conv2d(3, c1, kernel_size=11, stride=4, padding=2), nn.ReLU(inplace=True),
            nn.MaxPool2d(kernel_size=3, stride=2),
            nn.Conv2d(c1, c2, kernel_size=5, padding=2), nn.ReLU(inplace=True),
            nn.MaxPool2d(kernel_size=3, stride=2),
            nn.Conv2d(c2, c3, kernel_size=3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(c3, c4, kernel_size=3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(c4, c5, kernel_size=3, padding=1), nn.ReLU(inplace=True),
            nn.MaxPool2d(kernel_size=3, stride=2))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.classifier = nn.Sequential(
            nn.Dropout(), nn.Linear(c5, FC), nn.ReLU(inplace=True),
            nn.Dropout(), nn.Linear(FC, FC), nn.ReLU(inplace=True),
            nn.Linear(FC, num_classes))


def det_params(seed=15, num_classes=NCLS0):
    """kaiming-scaled weights, small non-zero biases, in SmallAlexNet().parameters() order."""
    gen = np.random.RandomState(seed)
    out = []
    for p in SmallAlexNet(num_classes).parameters():
        shp = tuple(p.shape)
        if len(shp) > 1:
            fan_in = int(np.prod(shp[1:]))
            out.append((gen.standard_normal(shp) * (2.0 / fan_in) ** 0.5).astype(np.float32))
        else:
            out.append((gen.uniform(-0.1, 0.1, shp)).astype(np.float32))
    return out


def load_params(model, arrays):
    with torch.no_grad():
        for p, a in zip(model.parameters(), arrays):
            p.copy_(torch.from_numpy(a))
    return model


def batches(seed=151, steps=4):
    """[(x [BATCH,3,67,67], y [BATCH] in 0..3)] — class-dependent mean so that losses move."""
    gen = np.random.RandomState(seed)
    out = []
    for _ in range(steps):
        y = gen.randint(0, 4, BATCH)
        x = gen.standard_normal((BATCH, 3, HW, HW)).astype(np.float32) + (y[:, None, None, None] - 1.5).astype(np.float32) * 0.5
        out.append((x.astype(np.float32), y.astype(np.int64)))
    return out
