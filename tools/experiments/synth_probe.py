"""Experiment (CPU, plain torch): how fast does small_VGG9 with torchvision's init (N(0, 0.01) classifier) learn a synthetic
task of a given design, and where does its accuracy saturate?  Used to pick the generator of
clsurvey_amd.data.synthetic_task (kind='blobs') so that the 10-task sweep of bench.py exercises the framework.

  python tools/experiments/synth_probe.py hw n_train epochs g amp noise_lr noise_px [lr]
"""
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

CFG = [64, "M", 64, "M", 64, 64, "M", 128, 128, "M"]


def make_net(hw, ncls, kaiming_fc=False):
    layers, c = [], 3
    for v in CFG:
        if v == "M":
            layers.append(nn.MaxPool2d(2, 2))
        else:
            layers += [nn.Conv2d(c, v, 3, padding=1), nn.ReLU(True)]
            c = v
    feat = nn.Sequential(*layers)
    d = c * (hw // 16) ** 2
    cls = nn.Sequential(nn.Linear(d, 128), nn.ReLU(True), nn.Linear(128, 128), nn.ReLU(True), nn.Linear(128, ncls))
    m = nn.Sequential(feat, nn.Flatten(), cls)
    for mod in m.modules():
        if isinstance(mod, nn.Conv2d):
            nn.init.kaiming_normal_(mod.weight, mode="fan_out", nonlinearity="relu")
            nn.init.constant_(mod.bias, 0)
        elif isinstance(mod, nn.Linear):
            if kaiming_fc:
                nn.init.kaiming_normal_(mod.weight, nonlinearity="relu")
            else:
                nn.init.normal_(mod.weight, 0, 0.01)
            nn.init.constant_(mod.bias, 0)
    return m


Q = [1.0]


def blobs(n, ncls, hw, g, amp, noise_lr, noise_px, gen, protos=None):
    if protos is None:
        protos = torch.randn((ncls, 3, g, g), generator=gen) * amp
    y = torch.randint(0, ncls, (n,), generator=gen)
    # overlapping classes: with probability 1 - q the image shows the prototype of a uniformly drawn class instead of its own
    other = torch.randint(0, ncls, (n,), generator=gen)
    z = torch.where(torch.rand((n,), generator=gen) < Q[0], y, other)
    lr = protos[z] + noise_lr * torch.randn((n, 3, g, g), generator=gen)
    x = F.interpolate(lr, size=(hw, hw), mode="nearest") + noise_px * torch.randn((n, 3, hw, hw), generator=gen)
    return x, y, protos


def main():
    a = sys.argv[1:]
    hw, ntr, epochs, g = int(a[0]), int(a[1]), int(a[2]), int(a[3])
    amp, nlr, npx = float(a[4]), float(a[5]), float(a[6])
    lr = float(a[7]) if len(a) > 7 else 1e-2
    kaiming = len(a) > 8 and a[8] == "k"
    Q[0] = float(a[9]) if len(a) > 9 else 1.0
    gen = torch.Generator().manual_seed(7001)
    xtr, ytr, protos = blobs(ntr, 20, hw, g, amp, nlr, npx, gen)
    xva, yva, _ = blobs(ntr // 4, 20, hw, g, amp, nlr, npx, gen, protos)
    # Bayes rule on the low-resolution block means (pixel noise averaged over a block is negligible for large blocks)
    with torch.no_grad():
        lrv = F.adaptive_avg_pool2d(xva, g)
        d = ((lrv[:, None] - protos[None]) ** 2).flatten(2).sum(2)
        print("template-matching accuracy on block means: %.3f" % float((d.argmin(1) == yva).float().mean()), flush=True)
    torch.manual_seed(0)
    m = make_net(hw, 20, kaiming)
    opt = torch.optim.SGD(m.parameters(), lr=lr, momentum=0.9)
    bs = 200
    for ep in range(epochs):
        t0 = time.time()
        perm = torch.randperm(ntr)
        m.train()
        tot = 0.0
        for i in range(0, ntr, bs):
            idx = perm[i:i + bs]
            opt.zero_grad()
            loss = F.cross_entropy(m(xtr[idx]), ytr[idx])
            loss.backward()
            opt.step()
            tot += float(loss) * len(idx)
        m.eval()
        with torch.no_grad():
            acc = sum(int((m(xva[i:i + 500]).argmax(1) == yva[i:i + 500]).sum()) for i in range(0, len(xva), 500)) / len(xva)
        print("epoch %d loss %.4f val %.3f (%.0fs)" % (ep, tot / ntr, acc, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
