"""G14: LwF fixture from the reference's UNCHANGED methods/LwF/{main_LWF,AlexNet_LwF}.py (dev container only).

  * distillation_loss (main_LWF.py:47-76): value and gradient for random student / teacher logits, T = 2 and T = 1.5;
  * AlexNet_LwF.forward on a tiny VGGSlim with two extra stacked heads: the list of head outputs;
  * one LwF objective (CrossEntropy on the last head + lambda * sum of the distillation terms) with all parameter
    gradients, as train_model_lwf builds it (main_LWF.py:184-202).
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
import models.VGGSlim as V  # noqa: E402
import methods.LwF.main_LWF as LW  # noqa: E402
from methods.LwF.AlexNet_LwF import AlexNet_LwF  # noqa: E402
from oracle import vgg_ref  # noqa: E402

TINY = [16, "M", 16, "M", 32, 32, "M", 32, 32, "M"]
V.cfg["tiny_VGG9"] = TINY


def main():
    out = {}
    gen = np.random.RandomState(14)
    for tag, T, n, c in (("a", 2.0, 6, 8), ("b", 1.5, 5, 20)):
        y = torch.from_numpy((gen.standard_normal((n, c)) * 2).astype(np.float32)).requires_grad_(True)
        t = torch.from_numpy((gen.standard_normal((n, c)) * 2).astype(np.float32))
        loss = LW.distillation_loss(y, t, T, c)
        loss.backward()
        out["d%s_y" % tag], out["d%s_t" % tag] = y.detach().numpy().copy(), t.numpy().copy()
        out["d%s_T" % tag] = np.array(T)
        out["d%s_loss" % tag], out["d%s_grad" % tag] = loss.detach().numpy().copy(), y.grad.numpy().copy()
    # wrapper with three heads (4, 8, 4 classes)
    m = V.VGGSlim(config="tiny_VGG9", num_classes=4, classifier_inputdim=32 * 2 * 2, classifier_dim1=24, classifier_dim2=24)
    params = vgg_ref.init_params(TINY, (24, 24), 4, 32, np.random.RandomState(141))
    with torch.no_grad():
        for p, q in zip(m.parameters(), params):
            p.copy_(q)
        for mod in m.classifier:
            if isinstance(mod, nn.Linear):
                mod.weight.mul_(20.0)
    w = AlexNet_LwF(m, last_layer_name=4)
    for i, nc in enumerate((8, 4)):
        h = nn.Linear(24, nc)
        with torch.no_grad():
            h.weight.copy_(torch.from_numpy((gen.standard_normal((nc, 24)) * 0.2).astype(np.float32)))
            h.bias.copy_(torch.from_numpy((gen.standard_normal(nc) * 0.1).astype(np.float32)))
        w.model.classifier.add_module(str(5 + i), h)
        out["head%d_w" % (i + 1)], out["head%d_b" % (i + 1)] = h.weight.detach().numpy().copy(), h.bias.detach().numpy().copy()
    x = torch.from_numpy(gen.standard_normal((6, 3, 32, 32)).astype(np.float32))
    y = torch.from_numpy(gen.randint(0, 4, size=(6,)).astype(np.int64))
    teacher = [torch.from_numpy((gen.standard_normal((6, nc)) * 2).astype(np.float32)) for nc in (4, 8)]
    out["x"], out["y"] = x.numpy().copy(), y.numpy().copy()
    out["teacher0"], out["teacher1"] = teacher[0].numpy().copy(), teacher[1].numpy().copy()
    outs = w(x)
    for i, o in enumerate(outs):
        out["out%d" % i] = o.detach().numpy().copy()
    lam, T = 10.0, 2.0
    task_loss = nn.CrossEntropyLoss()(outs[-1], y)
    dist = 0
    for idx in range(2):
        dist = dist + LW.distillation_loss(outs[idx], teacher[idx], T, teacher[idx].size(-1))
    total = lam * dist + task_loss
    w.zero_grad()
    total.backward()
    out["task_loss"], out["dist_loss"] = task_loss.detach().numpy().copy(), (lam * dist).detach().numpy().copy()
    out["param_names"] = np.array([n for n, _ in w.named_parameters()])
    for j, (n, p) in enumerate(w.named_parameters()):
        out["g%d" % j] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "G14_lwf.npz"), **out)
    print("wrote G14_lwf.npz", os.path.getsize(os.path.join(HERE, "G14_lwf.npz")) // 1024, "KiB", list(out["param_names"])[-6:])


if __name__ == "__main__":
    main()
