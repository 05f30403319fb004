"""G20: config 5 at its real widths (wide_VGG9_cl_512_512), from the reference's UNCHANGED code, dev container only.

  hat64      vgg_hat.Net.forward + Appr.criterion + backward + HAT_SGD.step + clamp, two batches of 8 at 3x64x64
  hat224     and of 4 at 3x224x224, the iNaturalist geometry of BASELINE configs[4]
             (methods/HAT/networks/vgg_hat.py:83-127, approaches/hat.py, HAT_utils.py)
  pack64     packnet Manager.do_batch x2 (forward, backward, make_grads_zero, PacknetSGD.step, make_pruned_zero), batches
  pack224    of 8 at 3x64x64 and of 4 at 3x224x224 (methods/packnet/main.py:164-198, prune.py:73-112)

Parameters, batches and owner masks are regenerated on both sides from seeds (g20_common.py); the fixture holds the
reference's outputs: logits / losses / gates in full, gradients and updated parameters at sampled positions with float64
checksums, and for PackNet the sha256 of every layer's zero bitmap (pruned positions are bit-exact).

    python tests/golden/make_g20.py        (about 2 minutes of CPU)
"""
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "harness"))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import harness  # noqa: E402

torch = harness.install()
import torch.nn as nn  # noqa: E402
import models.VGGSlim as V  # noqa: E402
import g20_common as C  # noqa: E402

torch.set_num_threads(16)
OUT = {}


def put(tag, a, seed):
    d = C.digest(a.detach().cpu().numpy() if hasattr(a, "detach") else a, seed)
    OUT[tag + "__v"], OUT[tag + "__s"] = d["v"], d["s"]


def raw_model(hw):
    return V.VGGSlim(config="wide_VGG9", num_classes=C.NCLS, classifier_inputdim=512 * (hw // 16) ** 2,
                     classifier_dim1=C.FC[0], classifier_dim2=C.FC[1])


def load(module, seed):
    named = [(n, tuple(p.shape)) for n, p in module.named_parameters()]
    with torch.no_grad():
        for (n, p), q in zip(module.named_parameters(), C.fill_params(named, seed)):
            p.copy_(torch.from_numpy(q))
    return [n for n, _ in named]


def hat(tag="hat64", hw=64, nb=8, seed=2001):
    """seeds: parameters `seed`, batches seed + 99 + step, digests seed + 199 + j / seed + 299 + j (hat64: 2001, 2100, 2200, 2300)"""
    import methods.HAT.networks.vgg_hat as VH
    import methods.HAT.approaches.hat as HA
    import methods.HAT.HAT_utils as HU
    torch.cuda.LongTensor = torch.LongTensor
    taskcla = [(0, C.NCLS), (1, C.NCLS), (2, C.NCLS)]
    net = VH.Net(raw_model(hw), (3, hw, hw), taskcla, uniform_init=True)
    names = load(net, seed)
    OUT[tag + "_param_names"] = np.array(names)
    smax, lamb, t, lr, mom, wd = 400.0, 0.75, 1, 0.05, 0.9, 1e-4
    OUT[tag + "_hyper"] = np.array([smax, lamb, t, lr, mom, wd])
    mask_pre, mask_back = HA.Appr.init_masks(t, net, smax)
    appr = HA.Appr.__new__(HA.Appr)
    appr.mask_pre, appr.lamb, appr.ce = mask_pre, lamb, nn.CrossEntropyLoss()
    opt = HU.HAT_SGD(net.parameters(), lr=lr, momentum=mom, weight_decay=wd)
    task = torch.LongTensor([t])
    net.train()
    for step, s in enumerate((3.1, 171.0)):
        x, y = (torch.from_numpy(a) for a in C.batch(seed + 99 + step, nb, hw))
        output, masks = net.forward(task, x, s=s)
        loss, reg = appr.criterion(output, y, masks)
        opt.zero_grad()
        loss.backward()
        OUT["%s_s%d_logits" % (tag, step)] = output.detach().numpy().copy()
        OUT["%s_s%d_loss" % (tag, step)] = np.array([float(loss), float(reg)])
        for i, mk in enumerate(masks):
            OUT["%s_s%d_mask%d" % (tag, step, i)] = mk.detach().numpy().copy()
        for j, (n, p) in enumerate(net.named_parameters()):
            if p.grad is not None:
                put("%s_s%d_grad_%s" % (tag, step, n), p.grad, seed + 199 + j)
        opt.step(net, mask_back, t, s, 50, smax, 10000)
        for n, p in net.named_parameters():
            if "embs" in n:
                p.data = torch.clamp(p.data, -6, 6)
        for j, (n, p) in enumerate(net.named_parameters()):
            put("%s_s%d_theta_%s" % (tag, step, n), p, seed + 299 + j)
        print(tag, "step", step, float(loss), float(reg))


def pack(tag, hw, nb, seed):
    import methods.packnet.main as PM
    import methods.packnet.networks as PN
    import methods.packnet.prune as PP
    from methods.packnet.packnetSGD import PacknetSGD
    raw = raw_model(hw)
    wrapper = PN.ModifiedWrapperModel(raw, 4, (3, hw, hw))
    wrapper.add_dataset("t1", C.NCLS)
    wrapper.add_dataset("t2", C.NCLS)
    wrapper.set_dataset("t2")
    names = load(wrapper, seed)
    OUT[tag + "_param_names"] = np.array(names)
    masks, layout = {}, []
    for i, mod in enumerate(wrapper.shared.modules()):
        if isinstance(mod, (nn.Conv2d, nn.Linear)):
            masks[i] = torch.from_numpy(C.owner_mask(seed + 100 + i, mod.weight.shape))
            layout.append(i)
    OUT[tag + "_layout"] = np.array(layout)
    lr, mom, wd = 0.01, 0.9, 5e-4
    OUT[tag + "_hyper"] = np.array([lr, mom, wd, nb, hw])
    mgr = PM.Manager.__new__(PM.Manager)
    mgr.args = SimpleNamespace(disable_pruning_mask=False)
    mgr.cuda, mgr.model, mgr.criterion = False, wrapper, nn.CrossEntropyLoss()
    mgr.pruner = PP.SparsePruner(wrapper, 0.5, masks, False, False, 2)
    mgr.pruner.current_masks = masks         # the state prune() leaves for the post-prune epochs (prune.py:43-71):
    mgr.pruner.make_pruned_zero()            # owner-0 weights are zero and stay zero, owner-1 weights are frozen
    opt = PacknetSGD(wrapper.parameters(), lr=lr, momentum=mom, weight_decay=wd)
    wrapper.train()
    for step in range(2):
        x, y = (torch.from_numpy(a) for a in C.batch(seed + 10 + step, nb, hw))
        meter = mgr.do_batch(opt, x, y)
        OUT["%s_s%d_err" % (tag, step)] = np.array(meter.value())
        for j, (n, p) in enumerate(wrapper.named_parameters()):
            put("%s_s%d_theta_%s" % (tag, step, n), p, seed + 300 + j)
            if p.dim() > 1 and n.startswith("shared"):
                OUT["%s_s%d_zeros_%s" % (tag, step, n)] = np.array(C.zero_pattern(p.detach().numpy()))
        print(tag, "step", step, meter.value())


if __name__ == "__main__":
    hat("hat64", 64, 8, 2001)
    hat("hat224", 224, 4, 7001)
    pack("pack64", 64, 8, 3000)
    pack("pack224", 224, 4, 4000)
    path = os.path.join(HERE, "G20_wide_widths.npz")
    np.savez_compressed(path, **OUT)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
