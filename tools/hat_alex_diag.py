#!/usr/bin/env python
"""Diagnostic: HAT-AlexNet step of the product against the same computation written with torch autograd on the device."""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import g20_common as C
from clsurvey_amd import models
from clsurvey_amd.methods import hat as HT
DEV = "cuda"
t, nb, s, smax, lamb = 1, 4, 3.1, 400.0, 0.75
net = HT.HatNetAlexnet(models.AlexNet(num_classes=C.NCLS), (3, 224, 224), [(0, C.NCLS), (1, C.NCLS), (2, C.NCLS)])
named = [(n, tuple(p.shape)) for n, p in net.named_parameters()]
with torch.no_grad():
    for (n, p), q in zip(net.named_parameters(), C.fill_params(named, 5001)):
        p.copy_(torch.from_numpy(q))
hat = HT.HatEngine(net, nb, (3, 224, 224), DEV)
mask_pre, mask_back = HT.init_masks(hat, t, smax)
hat.view.train()
gen = np.random.RandomState(5200)
masks = [torch.from_numpy((gen.rand(nb, d) < 0.5).astype(np.float32) * 2.0).to(DEV) for d in (256 * 6 * 6, 4096)]
hat.engine.auto_dropout = False
for li, m in zip(sorted(hat.engine.drops), masks):
    hat.engine.set_dropout(li, m)
x, y = (torch.from_numpy(a).to(DEV) for a in C.batch(5101, nb, 224))
ce, reg, logits = hat.step(t, x, y, s, mask_pre, lamb, None, True, want_logits=True)
mine = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
# ---- the same in torch autograd (fp64 on the device)
P = {n: p.detach().double().clone().requires_grad_(True) for n, p in net.named_parameters()}
tt = torch.tensor([t], device=DEV)
gates = [torch.sigmoid(s * P["conv_embs.%d.weight" % i][t]) for i in range(5)] + [torch.sigmoid(s * P["fc_embs.%d.weight" % i][t]) for i in range(2)]
h = x.double()
geo = [(4, 2), (1, 2), (1, 1), (1, 1), (1, 1)]
for i in range(5):
    h = F.relu(F.conv2d(h, P["convs.%d.weight" % i], P["convs.%d.bias" % i], stride=geo[i][0], padding=geo[i][1]))
    if i in (0, 1, 4):
        h = F.max_pool2d(h, 3, 2)
    h = h * gates[i].view(1, -1, 1, 1)
h = h.reshape(nb, -1)
for i in range(2):
    h = F.relu(F.linear(h * masks[i].double(), P["fcs.%d.weight" % i], P["fcs.%d.bias" % i])) * gates[5 + i]
out = F.linear(h, P["classifier.0.weight"], P["classifier.0.bias"])
regn, cnt = 0.0, 0.0
for g_, mp in zip(gates, mask_pre):
    aux = 1 - mp.double()
    regn = regn + (g_ * aux).sum(); cnt = cnt + aux.sum()
loss = F.cross_entropy(out, y) + lamb * regn / cnt
loss.backward()
print("loss mine %.6f torch %.6f" % (float(ce) + float(reg), float(loss)))
for n in mine:
    a, b = mine[n].double(), P[n].grad
    print("%-22s max %.2e l2 %.2e" % (n, float((a - b).abs().max() / b.abs().max()), float((a - b).norm() / b.norm())))
