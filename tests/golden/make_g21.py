"""G21: HAT on AlexNet (methods/HAT/networks/alexnet_hat.py:4-13 over vgg_hat.Net) from the reference's UNCHANGED code, dev
container only.  torchvision-shaped AlexNet at 3x224x224 (the net asserts a 6x6 feature map), 3 tasks x 20 classes:

  eval     forward at s = smax in eval mode (Dropout off): logits + gates
  train    one step in train mode with INJECTED Dropout masks (the net's drop_fc module is swapped for one that applies a
           preset 0 / 2 mask per call — masks are data): forward, Appr.criterion, backward, HAT_SGD.step, clamp

Parameters / batches / masks are regenerated on both sides from seeds (g20_common.py); the fixture holds the reference's
outputs (logits, loss, gates in full; gradients and updated parameters sampled + float64 checksums).
    python tests/golden/make_g21.py        (about 1 minute of CPU)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "harness"))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import harness  # noqa: E402

torch = harness.install()
import torch.nn as nn  # noqa: E402
import g20_common as C  # noqa: E402
from clsurvey_amd import models as M  # noqa: E402  (torchvision-shaped AlexNet module tree; standard torch layers only)

torch.set_num_threads(16)
OUT = {}
NB = 4


def put(tag, a, seed):
    d = C.digest(a.detach().cpu().numpy(), seed)
    OUT[tag + "__v"], OUT[tag + "__s"] = d["v"], d["s"]


class PresetDrop(nn.Module):
    """nn.Dropout(0.5) with the Bernoulli draw replaced by preset masks (values 0 or 2), one per call in order."""

    def __init__(self, masks):
        super().__init__()
        self.masks, self.i = masks, 0

    def forward(self, x):
        if not self.training:
            return x
        m = self.masks[self.i % len(self.masks)]
        self.i += 1
        return x * m


def drop_masks(seed, nb):
    gen = np.random.RandomState(seed)
    return [torch.from_numpy((gen.rand(nb, d) < 0.5).astype(np.float32) * 2.0) for d in (256 * 6 * 6, 4096)]


if __name__ == "__main__":
    import methods.HAT.networks.alexnet_hat as AH
    import methods.HAT.approaches.hat as HA
    import methods.HAT.HAT_utils as HU
    torch.cuda.LongTensor = torch.LongTensor
    taskcla = [(0, C.NCLS), (1, C.NCLS), (2, C.NCLS)]
    raw = M.AlexNet(num_classes=C.NCLS)
    net = AH.Net(raw, (3, 224, 224), taskcla)
    named = [(n, tuple(p.shape)) for n, p in net.named_parameters()]
    with torch.no_grad():
        for (n, p), q in zip(net.named_parameters(), C.fill_params(named, 5001)):
            p.copy_(torch.from_numpy(q))
    OUT["param_names"] = np.array([n for n, _ in named])
    smax, lamb, t, lr, mom, wd = 400.0, 0.75, 1, 0.05, 0.9, 1e-4
    OUT["hyper"] = np.array([smax, lamb, t, lr, mom, wd])
    task = torch.LongTensor([t])
    # ---- eval
    net.eval()
    x, y = (torch.from_numpy(a) for a in C.batch(5100, NB, 224))
    with torch.no_grad():
        out, masks = net.forward(task, x, s=smax)
    OUT["eval_logits"] = out.numpy().copy()
    for i, mk in enumerate(masks):
        OUT["eval_mask%d" % i] = mk.numpy().copy()
    # ---- one training step with injected masks
    mask_pre, mask_back = HA.Appr.init_masks(t, net, smax)
    appr = HA.Appr.__new__(HA.Appr)
    appr.mask_pre, appr.lamb, appr.ce = mask_pre, lamb, nn.CrossEntropyLoss()
    opt = HU.HAT_SGD(net.parameters(), lr=lr, momentum=mom, weight_decay=wd)
    net.drop_fc = PresetDrop(drop_masks(5200, NB))
    net.train()
    s = 3.1
    x, y = (torch.from_numpy(a) for a in C.batch(5101, NB, 224))
    out, masks = net.forward(task, x, s=s)
    loss, reg = appr.criterion(out, y, masks)
    opt.zero_grad()
    loss.backward()
    OUT["train_logits"] = out.detach().numpy().copy()
    OUT["train_loss"] = np.array([float(loss), float(reg), s])
    for j, (n, p) in enumerate(net.named_parameters()):
        if p.grad is not None:
            put("train_grad_" + n, p.grad, 5300 + j)
    opt.step(net, mask_back, t, s, 50, smax, 10000)
    for n, p in net.named_parameters():
        if "embs" in n:
            p.data = torch.clamp(p.data, -6, 6)
    for j, (n, p) in enumerate(net.named_parameters()):
        put("train_theta_" + n, p, 5400 + j)
    print("loss", float(loss), float(reg), "enable_warmup", net.enable_warmup, "smid", net.smid)
    path = os.path.join(HERE, "G21_hat_alexnet.npz")
    np.savez_compressed(path, **OUT)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
