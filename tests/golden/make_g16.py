"""G16: EBLL fixture from the reference's UNCHANGED methods/EBLL/{AlexNet_EBLL,Finetune_SGD_EBLL}.py (dev container only).

  * stage 1 (autoencoder): AlexNet_ENCODER on a tiny VGGSlim: forward (head output, autoencoder input, reconstruction),
    total = alpha * MSELoss(reconstruction, input) + CrossEntropy, gradients of the four autoencoder tensors, and the
    autoencoder after three optim.Adadelta steps on that batch (Finetune_SGD_EBLL.py:139-160, 497);
  * stage 2: AlexNet_EBLL with two earlier encoders and heads (4, 8) + a new head (4): outputs and codes, the objective
    total = reg_lambda * sum distillation + CrossEntropy + reg_alpha * sum MSELoss(code, target code) as train_model_ebll
    builds it (Finetune_SGD_EBLL.py:297-317), with the gradients of every feature-extractor / classifier parameter.
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
import torch.optim as optim  # noqa: E402
import models.VGGSlim as V  # noqa: E402
import methods.EBLL.Finetune_SGD_EBLL as EB  # noqa: E402
from methods.EBLL.AlexNet_EBLL import AlexNet_ENCODER, AlexNet_EBLL, AutoEncoder  # noqa: E402
from oracle import vgg_ref  # noqa: E402

TINY = [16, "M", 16, "M", 32, 32, "M", 32, 32, "M"]
V.cfg["tiny_VGG9"] = TINY
F_DIM = 32 * 2 * 2


def tiny_net(seed):
    m = V.VGGSlim(config="tiny_VGG9", num_classes=4, classifier_inputdim=F_DIM, classifier_dim1=24, classifier_dim2=24)
    params = vgg_ref.init_params(TINY, (24, 24), 4, 32, np.random.RandomState(seed))
    with torch.no_grad():
        for p, q in zip(m.parameters(), params):
            p.copy_(q)
        for mod in m.classifier:
            if isinstance(mod, nn.Linear):
                mod.weight.mul_(20.0)
    return m


def set_ae(ae, gen, out, tag):
    with torch.no_grad():
        for name, p in ae.named_parameters():
            v = (gen.standard_normal(tuple(p.shape)) * (0.3 if p.dim() > 1 else 0.05)).astype(np.float32)
            p.copy_(torch.from_numpy(v))
            out["%s_%s" % (tag, name)] = v.copy()


def main():
    out = {}
    gen = np.random.RandomState(16)
    x = torch.from_numpy(gen.standard_normal((6, 3, 32, 32)).astype(np.float32))
    y = torch.from_numpy(gen.randint(0, 4, size=(6,)).astype(np.int64))
    out["x"], out["y"] = x.numpy().copy(), y.numpy().copy()

    # ---------------- stage 1
    alpha, dim = 0.1, 10
    enc = AlexNet_ENCODER(tiny_net(161), dim=dim, last_layer_name=4, num_ftrs=F_DIM)
    set_ae(enc.autoencoder, gen, out, "s1")
    opt = optim.Adadelta(enc.autoencoder.parameters(), 0.01)
    for step in range(3):
        opt.zero_grad()
        enc.zero_grad()
        outputs, e_in, e_out = enc(x)
        task_loss = nn.CrossEntropyLoss()(outputs, y)
        e_loss = nn.MSELoss()(e_out, e_in)
        total = alpha * e_loss + task_loss
        total.backward()
        if step == 0:
            out["s1_out"], out["s1_in"], out["s1_recon"] = (t.detach().numpy().copy() for t in (outputs, e_in, e_out))
            out["s1_task_loss"], out["s1_enc_loss"] = task_loss.detach().numpy().copy(), e_loss.detach().numpy().copy()
            for name, p in enc.autoencoder.named_parameters():
                out["s1_grad_%s" % name] = p.grad.numpy().copy()
        opt.step()
    for name, p in enc.autoencoder.named_parameters():
        out["s1_after3_%s" % name] = p.detach().numpy().copy()
    out["s1_alpha"] = np.array(alpha)

    # ---------------- stage 2
    lam, reg_alpha, T = 10.0, 2.0, 2
    base = tiny_net(162)
    ae0, ae1 = AutoEncoder(F_DIM, 10), AutoEncoder(F_DIM, 6)
    set_ae(ae0, gen, out, "s2_ae0")
    set_ae(ae1, gen, out, "s2_ae1")
    w = AlexNet_EBLL(base, ae0, last_layer_name=4)
    w.autoencoders.add_module("1", ae1.encode)
    for i, nc in enumerate((8, 4)):
        h = nn.Linear(24, nc)
        with torch.no_grad():
            h.weight.copy_(torch.from_numpy((gen.standard_normal((nc, 24)) * 0.2).astype(np.float32)))
            h.bias.copy_(torch.from_numpy((gen.standard_normal(nc) * 0.1).astype(np.float32)))
        w.classifier.add_module(str(5 + i), h)
        out["s2_head%d_w" % (i + 1)], out["s2_head%d_b" % (i + 1)] = h.weight.detach().numpy().copy(), h.bias.detach().numpy().copy()
    target_logits = [torch.from_numpy((gen.standard_normal((6, nc)) * 2).astype(np.float32)) for nc in (4, 8)]
    target_codes = [torch.from_numpy(gen.uniform(0.1, 0.9, (6, d)).astype(np.float32)) for d in (10, 6)]
    for i in range(2):
        out["s2_tlogits%d" % i], out["s2_tcodes%d" % i] = target_logits[i].numpy().copy(), target_codes[i].numpy().copy()
    outs, codes = w(x)
    for i, o in enumerate(outs):
        out["s2_out%d" % i] = o.detach().numpy().copy()
    for i, c in enumerate(codes):
        out["s2_code%d" % i] = c.detach().numpy().copy()
    task_loss = nn.CrossEntropyLoss()(outs[-1], y)
    dist = 0.0
    code_loss = 0.0
    for idx in range(2):
        dist = dist + EB.distillation_loss(outs[idx], target_logits[idx], T, target_logits[idx].size(-1))
    for idx in range(2):
        code_loss = code_loss + nn.MSELoss()(codes[idx], target_codes[idx])
    total = lam * dist + task_loss + reg_alpha * code_loss
    w.zero_grad()
    total.backward()
    out["s2_task_loss"], out["s2_dist_loss"], out["s2_code_loss"] = (task_loss.detach().numpy().copy(),
                                                                     (lam * dist).detach().numpy().copy(),
                                                                     code_loss.detach().numpy().copy())
    out["s2_lambda"], out["s2_reg_alpha"] = np.array(lam), np.array(reg_alpha)
    names = [n for n, p in w.named_parameters() if not n.startswith("autoencoders")]
    out["s2_param_names"] = np.array(names)
    for j, n in enumerate(names):
        out["s2_g%d" % j] = dict(w.named_parameters())[n].grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "G16_ebll.npz"), **out)
    print("wrote G16_ebll.npz", os.path.getsize(os.path.join(HERE, "G16_ebll.npz")) // 1024, "KiB", names[-6:],
          float(task_loss), float(lam * dist), float(code_loss))


if __name__ == "__main__":
    main()
