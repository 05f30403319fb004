"""G13: IMM fixture from the reference's UNCHANGED methods/IMM/{merge,main_L2transfer}.py (dev container only).

Three tiny VGGSlim task models with deterministic weights:
  * main_L2transfer.update_reg_params after a head swap (which parameters get entries, omega = 1, init_val);
  * merge.IMM_merge_models in mean and mode form (precisions = deterministic positive tensors);
  * merge.diag_fisher on two phases of two batches with torch.multinomial replaced by argmax (the sampled labels are the
    only random input; the fixture pins the arithmetic: 1e-8 start, mean nll, grad^2 / #batches per phase, head excluded).
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
import methods.IMM.merge as MG  # noqa: E402
import methods.IMM.main_L2transfer as L2  # noqa: E402
from oracle import vgg_ref  # noqa: E402

TINY = [16, "M", 16, "M", 32, 32, "M", 32, 32, "M"]
V.cfg["tiny_VGG9"] = TINY
FC, NCLS, HW = (24, 24), 5, 32


def build(seed):
    m = V.VGGSlim(config="tiny_VGG9", num_classes=NCLS, classifier_inputdim=32 * 2 * 2, classifier_dim1=FC[0],
                  classifier_dim2=FC[1])
    params = vgg_ref.init_params(TINY, FC, NCLS, HW, np.random.RandomState(seed))
    with torch.no_grad():
        for p, q in zip(m.parameters(), params):
            p.copy_(q)
        for mod in m.classifier:
            if isinstance(mod, nn.Linear):
                mod.weight.mul_(20.0)            # N(0, .01) heads would make every Fisher entry ~1e-12
    return m


def main():
    out = {}
    gen = np.random.RandomState(13)
    models = [build(130 + i) for i in range(3)]
    names = [n for n, _ in models[0].named_parameters()]
    out["param_names"] = np.array(names)
    out["model_seeds"] = np.array([130, 131, 132])      # models are rebuilt in the tests: vgg_ref.init_params(seed), Linear weights x20
    KEEP = (0, 1, 6, 7, 12, 13)                          # tensors whose merges are stored (conv1, a mid conv, fc1)
    head = ["classifier.4.weight", "classifier.4.bias"]
    # ---- merge
    prec = [{n: torch.from_numpy(gen.uniform(1e-6, 1.0, size=tuple(p.shape)).astype(np.float32)) for n, p in m.named_parameters()
             if n not in head} for m in models]
    for i, pr in enumerate(prec):
        for j, n in enumerate(names):
            if n in pr and j in KEEP:
                out["prec%d_p%d" % (i, j)] = pr[n].numpy().copy()
    for idx in (1, 2):
        mean_m = MG.IMM_merge_models(models, idx, head, mean_mode=True)
        sump = {n: sum(prec[i][n] for i in range(idx + 1)) for n in prec[0]}
        s = None
        for i in range(idx + 1):                    # the reference's incremental sum order (merge.py:108-109)
            s = prec[i] if s is None else {n: p + prec[i][n] for n, p in s.items()}
        mode_m = MG.IMM_merge_models(models, idx, head, precision=prec, sum_precision=s, mean_mode=False)
        for j, (pm, pd) in enumerate(zip(mean_m.parameters(), mode_m.parameters())):
            if j in KEEP or j >= 16:                     # + the (unmerged) head
                out["mean%d_p%d" % (idx, j)] = pm.detach().numpy().copy()
                out["mode%d_p%d" % (idx, j)] = pd.detach().numpy().copy()
    # ---- diag_fisher with argmax "sampling"
    torch.multinomial = lambda probs, n, *a, **k: probs.argmax(1, keepdim=True)
    xs = {"train": [torch.from_numpy(gen.standard_normal((6, 3, HW, HW)).astype(np.float32)) for _ in range(2)],
          "val": [torch.from_numpy(gen.standard_normal((4, 3, HW, HW)).astype(np.float32)) for _ in range(3)]}
    for ph, lst in xs.items():
        for b, x in enumerate(lst):
            out["fx_%s%d" % (ph, b)] = x.numpy().copy()
    m = models[0]
    m.params = {n: p for n, p in m.named_parameters() if p.requires_grad}
    dataset = {ph: [(x, torch.zeros(x.shape[0], dtype=torch.int64)) for x in lst] for ph, lst in xs.items()}
    fisher = MG.diag_fisher(m, dataset, exclude_params=head)
    assert sorted(fisher) == sorted(n for n in names if n not in head)
    for j, n in enumerate(names):
        if n in fisher:
            out["fisher_p%d" % j] = fisher[n].detach().numpy().copy()
    # ---- update_reg_params after a head swap (main_L2transfer.py:118-139)
    m2 = build(140)
    m2.reg_params = {p: {"omega": torch.zeros_like(p), "init_val": torch.zeros_like(p)} for p in m2.parameters()}
    last = len(m2.classifier) - 1
    m2.classifier._modules[str(last)] = nn.Linear(m2.classifier[last].in_features, 7)
    ps = list(m2.parameters())
    m2.reg_params.pop(ps[-1], None); m2.reg_params.pop(ps[-2], None)
    rp = L2.update_reg_params(m2)
    out["urp_count"] = np.array(len([p for p in m2.parameters() if p in rp]))
    out["urp_omega_all_ones"] = np.array(all(bool((rp[p]["omega"] == 1).all()) for p in m2.parameters()))
    out["urp_init_equals_theta"] = np.array(all(torch.equal(rp[p]["init_val"], p.data) for p in m2.parameters()))
    np.savez_compressed(os.path.join(HERE, "G13_imm.npz"), **out)
    print("wrote G13_imm.npz", os.path.getsize(os.path.join(HERE, "G13_imm.npz")) // 1024, "KiB", out["urp_count"],
          out["urp_omega_all_ones"], out["urp_init_equals_theta"])


if __name__ == "__main__":
    main()
