"""Generate golden fixtures by running the REAL reference (/root/reference) on CPU.

Run in the dev container only:   python tests/golden/make_golden.py
Outputs tests/golden/G*.npz (inputs + expected outputs; data only).
The reference is imported through tests/golden/harness (torchvision stub,
.cuda() -> identity).  Nothing here travels as code to the GPU box tests: the
tests read only the .npz files.

Fixture ids follow SURVEY.md §8c (G1..G9).
"""
import os
import sys
import copy

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "harness"))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import harness  # noqa: E402

torch = harness.install()
import torch.nn as nn  # noqa: E402
import models.VGGSlim as V  # noqa: E402  (reference)
import methods.EWC.main_EWC as EWC_main  # noqa: E402
import methods.EWC.train_EWC as EWC_train  # noqa: E402
import methods.MAS.train_MAS as MAS_train  # noqa: E402
import methods.SI.train_SI as SI_train  # noqa: E402
from oracle import vgg_ref  # noqa: E402  (only for deterministic init helper)

TINY = [16, "M", 16, "M", 32, 32, "M", 32, 32, "M"]
V.cfg["tiny_VGG9"] = TINY
torch.set_num_threads(4)


def np_(t):
    return t.detach().cpu().numpy().copy()


def build(cfg_name, cfg, fc, ncls, hw, seed):
    """reference VGGSlim with deterministic numpy-driven weights."""
    m = V.VGGSlim(config=cfg_name, num_classes=ncls,
                  classifier_inputdim=[v for v in cfg if v != "M"][-1] * (hw // 16) ** 2,
                  classifier_dim1=fc[0], classifier_dim2=fc[1])
    gen = np.random.RandomState(seed)
    params = vgg_ref.init_params(cfg, fc, ncls, hw, gen)
    with torch.no_grad():
        for p, q in zip(m.parameters(), params):
            p.copy_(q)
    return m, params


def data(gen, n, hw, ncls):
    x = torch.from_numpy(gen.standard_normal((n, 3, hw, hw)).astype(np.float32))
    y = torch.from_numpy(gen.randint(0, ncls, size=(n,)).astype(np.int64))
    return x, y


def save(name, **kw):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **kw)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


# ------------------------------------------------------------------ G1 fwd/bwd
def g1():
    out = {}
    # (a) tiny config, full tensors
    m, params = build("tiny_VGG9", TINY, (24, 24), 5, 32, seed=11)
    gen = np.random.RandomState(12)
    x, y = data(gen, 6, 32, 5)
    for kind in ("ce_mean", "ce_sum", "mse_sum_zero"):
        m.zero_grad()
        m.eval()
        logits = m(x)
        if kind == "ce_mean":
            loss = nn.CrossEntropyLoss()(logits, y)                       # train_EWC.py:183
        elif kind == "ce_sum":
            loss = torch.nn.functional.nll_loss(
                torch.nn.functional.log_softmax(logits, dim=1), y, size_average=False)  # main_EWC.py:148
        else:
            loss = torch.nn.MSELoss(size_average=False)(logits, torch.zeros(logits.size()))  # train_MAS.py:556
        loss.backward()
        out["tiny_%s_loss" % kind] = np_(loss)
        out["tiny_%s_logits" % kind] = np_(logits)
        for i, p in enumerate(m.parameters()):
            out["tiny_%s_g%d" % (kind, i)] = np_(p.grad)
    out["tiny_x"], out["tiny_y"] = np_(x), np_(y)
    for i, p in enumerate(params):
        out["tiny_p%d" % i] = np_(p)
    # (b) small_VGG9 full width @64x64: regenerated inputs, checksums of grads
    m, params = build("small_VGG9", vgg_ref.CFGS["small_VGG9"], (128, 128), 20, 64, seed=21)
    gen = np.random.RandomState(22)
    x, y = data(gen, 4, 64, 20)
    m.zero_grad()
    logits = m(x)
    loss = nn.CrossEntropyLoss()(logits, y)
    loss.backward()
    out["small_logits"], out["small_loss"] = np_(logits), np_(loss)
    for i, p in enumerate(m.parameters()):
        g = p.grad.double()
        out["small_g%d_stats" % i] = np.array([g.sum().item(), g.abs().sum().item(),
                                               g.pow(2).sum().sqrt().item()])
        out["small_g%d_head" % i] = np_(p.grad.flatten()[:64])
    save("G1_vgg_fwd_bwd", **out)


# ------------------------------------------------------------ G2 EWC Fisher
def g2():
    m, params = build("tiny_VGG9", TINY, (24, 24), 5, 32, seed=31)
    gen = np.random.RandomState(32)
    out = {}
    for i, p in enumerate(params):
        out["p%d" % i] = np_(p)
    for task in range(2):
        batches = [data(gen, 8, 32, 5) for _ in range(3)]
        for b, (x, y) in enumerate(batches):
            out["t%d_x%d" % (task, b)], out["t%d_y%d" % (task, b)] = np_(x), np_(y)
        # accumulate_EWC_weights body (main_EWC.py:109-120) without the DataLoader plumbing
        if not hasattr(m, "reg_params"):
            m.reg_params = EWC_main.initialize_reg_params(m)
        m.reg_params = EWC_main.store_prev_reg_params(m)
        m = EWC_main.diag_fisher(m, batches, 24)
        m.reg_params = EWC_main.accumelate_reg_params(m)
        for i, p in enumerate(m.parameters()):
            out["t%d_omega%d" % (task, i)] = np_(m.reg_params[p]["omega"])
    save("G2_ewc_fisher", **out)


# ------------------------------------------------------------ G3 MAS omega
def g3():
    m, params = build("tiny_VGG9", TINY, (24, 24), 5, 32, seed=41)
    gen = np.random.RandomState(42)
    out = {}
    for i, p in enumerate(params):
        out["p%d" % i] = np_(p)
    sizes = [8, 8, 5]  # short last batch quirk (train_MAS.py:563)
    batches = [data(gen, n, 32, 5) for n in sizes]
    for b, (x, y) in enumerate(batches):
        out["x%d" % b], out["y%d" % b] = np_(x), np_(y)
    m.reg_params = MAS_train.initialize_reg_params(m)
    opt = MAS_train.Objective_After_SGD(m.parameters(), lr=0.0001, momentum=0.9)
    m = MAS_train.compute_importance_l2(m, opt, lambda o, e, l: o, [batches], False)
    for i, p in enumerate(m.parameters()):
        out["omega%d" % i] = np_(m.reg_params[p]["omega"])
    save("G3_mas_omega", **out)


# ------------------------------------------- G4 SI steps + consolidation
def g4():
    out = {}
    for tag, wd in (("wd0", 0.0), ("wd1", 1e-4)):
        m, params = build("tiny_VGG9", TINY, (24, 24), 5, 32, seed=51)
        gen = np.random.RandomState(52)
        m.reg_params = SI_train.initialize_reg_params(m)
        # non-trivial omega / init_val so the penalty is exercised
        for p in m.parameters():
            rp = m.reg_params[p]
            rp["omega"] = torch.from_numpy((1e-2 * np.abs(gen.standard_normal(tuple(p.shape)))).astype(np.float32))
            rp["init_val"] = p.data.clone() + torch.from_numpy(
                (0.05 * gen.standard_normal(tuple(p.shape))).astype(np.float32))
        m.reg_params["lambda"] = 400
        for i, p in enumerate(m.parameters()):
            out["%s_p%d" % (tag, i)] = np_(p)
            out["%s_omega%d" % (tag, i)] = np_(m.reg_params[p]["omega"])
            out["%s_init%d" % (tag, i)] = np_(m.reg_params[p]["init_val"])
        opt = SI_train.Elastic_SGD(m.parameters(), 1e-2, momentum=0.9, weight_decay=wd)
        crit = nn.CrossEntropyLoss()
        m.train(True)
        for s in range(3):
            x, y = data(gen, 8, 32, 5)
            out["%s_x%d" % (tag, s)], out["%s_y%d" % (tag, s)] = np_(x), np_(y)
            opt.zero_grad()
            loss = crit(m(x), y)
            loss.backward()
            opt.step(m.reg_params)
            for i, p in enumerate(m.parameters() if s == 2 else []):
                out["%s_s%d_theta%d" % (tag, s, i)] = np_(p)
                out["%s_s%d_w%d" % (tag, s, i)] = np_(m.reg_params[p]["w"])
                out["%s_s%d_buf%d" % (tag, s, i)] = np_(opt.state[p]["momentum_buffer"])
        lam = m.reg_params.pop("lambda")
        m.reg_params = SI_train.update_reg_params(m)
        for i, p in enumerate(m.parameters()):
            out["%s_cons_omega%d" % (tag, i)] = np_(m.reg_params[p]["omega"])
            out["%s_cons_init%d" % (tag, i)] = np_(m.reg_params[p]["init_val"])
            out["%s_cons_w%d" % (tag, i)] = np_(m.reg_params[p]["w"])
    save("G4_si", **out)


# ------------------------------------------- G5 penalised SGD (EWC & MAS classes)
def g5():
    out = {}
    for tag, cls, wd in (("ewc", EWC_train.Weight_Regularized_SGD, 0.0),
                         ("mas", MAS_train.Weight_Regularized_SGD, 5e-4)):
        m, params = build("tiny_VGG9", TINY, (24, 24), 5, 32, seed=61)
        gen = np.random.RandomState(62)
        m.reg_params = EWC_main.initialize_reg_params(m)
        plist = list(m.parameters())
        for p in plist:
            rp = m.reg_params[p]
            rp["omega"] = torch.from_numpy((1e-2 * np.abs(gen.standard_normal(tuple(p.shape)))).astype(np.float32))
            rp["init_val"] = p.data.clone() + torch.from_numpy(
                (0.05 * gen.standard_normal(tuple(p.shape))).astype(np.float32))
        # new head: the last two params are NOT in reg_params (main_EWC.py:52)
        del m.reg_params[plist[-1]], m.reg_params[plist[-2]]
        m.reg_params["lambda"] = 400 if tag == "ewc" else 3
        for i, p in enumerate(plist):
            out["%s_p%d" % (tag, i)] = np_(p)
            if p in m.reg_params:
                out["%s_omega%d" % (tag, i)] = np_(m.reg_params[p]["omega"])
                out["%s_init%d" % (tag, i)] = np_(m.reg_params[p]["init_val"])
        opt = cls(m.parameters(), 1e-2, momentum=0.9, weight_decay=wd)
        crit = nn.CrossEntropyLoss()
        m.train(True)
        for s in range(3):
            x, y = data(gen, 8, 32, 5)
            out["%s_x%d" % (tag, s)], out["%s_y%d" % (tag, s)] = np_(x), np_(y)
            opt.zero_grad()
            loss = crit(m(x), y)
            loss.backward()
            opt.step(m.reg_params)
            out["%s_s%d_loss" % (tag, s)] = np_(loss)
            for i, p in enumerate(plist if s == 2 else []):
                out["%s_s%d_theta%d" % (tag, s, i)] = np_(p)
                out["%s_s%d_buf%d" % (tag, s, i)] = np_(opt.state[p]["momentum_buffer"])
    save("G5_reg_sgd", **out)


# ------------------------------------------- G9 schedule traces
def g9():
    out = {}
    gen = np.random.RandomState(91)
    for tag, mod in (("ewc", EWC_train), ("si", SI_train)):
        for case in range(3):
            improved = gen.rand(40) < (0.15 + 0.2 * case)
            opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.01)
            lr, count, trace = 0.01, 0, []
            for ep, imp in enumerate(improved):
                opt, lr, cont = mod.set_lr(opt, lr, count)
                trace.append((ep, lr, float(cont)))
                if not cont:
                    break
                count = 0 if imp else count + 1
            out["%s_c%d_improved" % (tag, case)] = improved
            out["%s_c%d_trace" % (tag, case)] = np.array(trace, dtype=np.float64)
    # the reference's HyperparameterFramework.hyperparamDecay (framework_train.py:168-216): value of every
    # hyper-parameter after each of 9 consecutive failed attempts, for methods with 1 / 2 / 3 hyper-parameters and for a
    # method that brings its own decay_operator
    import collections
    import operator
    import types
    import framework.framework_train as FT
    cases = {"one": ([("lambda", 400.0)], None), "two": ([("smax", 800.0), ("c", 2.5)], None),
             "three": ([("a", 8.0), ("b", 4.0), ("c", 2.0)], None), "sub": ([("margin", 1.0), ("k", 3.0)], operator.sub)}
    for tag, (hp, op) in cases.items():
        method = types.SimpleNamespace(hyperparams=collections.OrderedDict(hp))
        if op is not None:
            method.decay_operator = op
        hf = FT.HyperparameterFramework(method)
        args = types.SimpleNamespace(decaying_factor=0.5)
        manager = types.SimpleNamespace(method=method)
        rows = [[float(v) for v in hf.hyperparams.values()]]
        for _ in range(9):
            hf.hyperparamDecay(args, manager)
            rows.append([float(v) for v in hf.hyperparams.values()])
        out["decay_%s" % tag] = np.array(rows, dtype=np.float64)
        out["decay_%s_keys" % tag] = np.array([k for k, _ in hp])
    save("G9_schedules", **out)




# ------------------------------------------- G7 PackNet masks (bit-exact)
def g7():
    import methods.packnet.prune as PP
    import methods.packnet.packnetSGD as PS
    gen = np.random.RandomState(71)

    class M(nn.Module):
        def __init__(self):
            super().__init__()
            self.shared = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 12, 3, padding=1),
                                        nn.ReLU(), nn.Linear(48, 40), nn.ReLU(), nn.Linear(40, 24))
    m = M()
    layers = [(i, mod) for i, mod in enumerate(m.shared.modules()) if isinstance(mod, (nn.Conv2d, nn.Linear))]
    with torch.no_grad():
        for _, mod in layers:
            mod.weight.copy_(torch.from_numpy(gen.standard_normal(tuple(mod.weight.shape)).astype(np.float32) * 0.1))
            mod.bias.copy_(torch.from_numpy(gen.standard_normal(tuple(mod.bias.shape)).astype(np.float32) * 0.1))
        # exact ties around the cutoff and exact zeros, to pin the <= / kthvalue semantics
        w0 = layers[0][1].weight
        w0.view(-1)[:20] = 0.05
        w0.view(-1)[20:30] = -0.05
        w0.view(-1)[30:34] = 0.0
    out = {"layer_idx": np.array([i for i, _ in layers])}
    masks = {i: torch.zeros(mod.weight.shape, dtype=torch.uint8) for i, mod in layers}

    def snap(tag):
        for i, mod in layers:
            out["%s_w%d" % (tag, i)] = np_(mod.weight)
            out["%s_b%d" % (tag, i)] = np_(mod.bias)
            if mod.weight.grad is not None:
                out["%s_g%d" % (tag, i)] = np_(mod.weight.grad)
                out["%s_gb%d" % (tag, i)] = np_(mod.bias.grad)

    def snap_masks(tag, md):
        for i, _ in layers:
            out["%s_m%d" % (tag, i)] = md[i].numpy().copy()

    snap("init")
    for task, perc in ((1, 0.75), (2, 0.5)):
        pr = PP.SparsePruner(m, perc, masks, False, False, task)
        pr.make_finetuning_mask()
        snap_masks("t%d_ft" % task, pr.current_masks)
        opt = PS.PacknetSGD(m.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
        for s in range(2):
            for _, mod in layers:
                mod.weight.grad = torch.from_numpy(gen.standard_normal(tuple(mod.weight.shape)).astype(np.float32))
                mod.bias.grad = torch.from_numpy(gen.standard_normal(tuple(mod.bias.shape)).astype(np.float32))
            for i, mod in layers:
                out["t%d_s%d_rawg%d" % (task, s, i)] = np_(mod.weight.grad)
                out["t%d_s%d_rawgb%d" % (task, s, i)] = np_(mod.bias.grad)
            pr.make_grads_zero()
            opt.step()
            pr.make_pruned_zero()
            snap("t%d_s%d" % (task, s))
        pr.current_masks = None
        pr.prune()
        snap("t%d_pruned" % task)
        snap_masks("t%d_pruned" % task, pr.current_masks)
        masks = pr.current_masks
    pr = PP.SparsePruner(m, 0.5, masks, False, False, 2)
    pr.apply_mask(1)
    snap("apply1")
    save("G7_packnet", **out)


# ------------------------------------------- G8 HAT gates / criterion / HAT_SGD
def g8():
    import methods.HAT.networks.vgg_hat as VH
    import methods.HAT.approaches.hat as HA
    import methods.HAT.HAT_utils as HU
    torch.cuda.LongTensor = torch.LongTensor
    raw, params = build("tiny_VGG9", TINY, (24, 24), 5, 32, seed=81)
    gen = np.random.RandomState(82)
    taskcla = [(0, 5), (1, 5), (2, 5)]
    net = VH.Net(raw, (3, 32, 32), taskcla, uniform_init=True)
    with torch.no_grad():
        for emb in list(net.conv_embs) + list(net.fc_embs):
            emb.weight.copy_(torch.from_numpy(gen.uniform(-1.5, 2.0, size=tuple(emb.weight.shape)).astype(np.float32)))
    out = {}
    names = [n for n, _ in net.named_parameters()]
    out["param_names"] = np.array(names)
    for n, p in net.named_parameters():
        out["p_" + n] = np_(p)
    smax, lamb, t = 50.0, 0.75, 1
    mask_pre, mask_back = HA.Appr.init_masks(t, net, smax)
    for i, mp in enumerate(mask_pre):
        out["mask_pre%d" % i] = np_(mp)
    for n, v in mask_back.items():
        out["mask_back_" + n] = np_(v)
    appr = HA.Appr.__new__(HA.Appr)
    appr.mask_pre, appr.lamb, appr.ce = mask_pre, lamb, torch.nn.CrossEntropyLoss()
    opt = HU.HAT_SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    task = torch.LongTensor([t])
    net.train()
    for step, s in enumerate((7.3, 23.0)):
        x, y = data(gen, 8, 32, 5)
        out["x%d" % step], out["y%d" % step] = np_(x), np_(y)
        output, masks = net.forward(task, x, s=s)
        loss, reg = appr.criterion(output, y, masks)
        opt.zero_grad()
        loss.backward()
        out["s%d_logits" % step], out["s%d_loss" % step], out["s%d_reg" % step] = np_(output), np_(loss), np_(reg)
        for i, mk in enumerate(masks):
            out["s%d_mask%d" % (step, i)] = np_(mk)
        for n, p in net.named_parameters():
            if p.grad is not None:
                out["s%d_rawgrad_%s" % (step, n)] = np_(p.grad)
        opt.step(net, mask_back, t, s, 50, smax, 10000)
        for n, p in net.named_parameters():
            if "embs" in n:
                p.data = torch.clamp(p.data, -6, 6)
        for n, p in net.named_parameters():
            out["s%d_theta_%s" % (step, n)] = np_(p)
            if p in opt.state and "momentum_buffer" in opt.state[p]:
                out["s%d_buf_%s" % (step, n)] = np_(opt.state[p]["momentum_buffer"])
    out["hyper"] = np.array([smax, lamb, t, 0.05, 0.9, 1e-4])
    save("G8_hat", **out)


# ------------------------------------------- G6 GEM gradient memory + projection
def g6():
    """store_grad / overwrite_grad / project2cone2 of the real gem.py. quadprog is NOT installed: the QP inside
    project2cone2 is solved by the scipy stand-in in tests/golden/harness/quadprog.py (SLSQP, ftol 1e-15), so
    the projected vectors pin gem.py's own arithmetic around the QP, not quadprog's."""
    import methods.rehearsal.model.gem as GEM
    gen = np.random.RandomState(61)
    out = {}
    shapes = [(4, 3, 3, 3), (4,), (6, 4), (6,)]
    params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
    dims = [p.numel() for p in params]
    G = torch.zeros(sum(dims), 5)
    for tid in (0, 2, 3):
        for pi, p in enumerate(params):
            p.grad = torch.from_numpy(gen.standard_normal(tuple(p.shape)).astype(np.float32))
            out["grad_t%d_%d" % (tid, pi)] = np_(p.grad)
        GEM.store_grad(lambda: params, G, dims, tid)
    out["G"] = np_(G)
    GEM.overwrite_grad(lambda: params, G[:, 2] * 2.0, dims)
    for i, p in enumerate(params):
        out["overwritten_%d" % i] = np_(p.grad)
    for case in range(6):
        t = [1, 2, 3, 5, 7, 9][case]
        P_ = 200 + 37 * case
        mem = torch.from_numpy(gen.standard_normal((P_, t)).astype(np.float32))
        g = torch.from_numpy(gen.standard_normal((P_, 1)).astype(np.float32))
        if case % 2 == 0:      # make sure some constraints are violated
            g = g - 0.5 * mem[:, :1]
        margin = [0.5, 1.0, 0.0, 1.0, 0.5, 1.0][case]
        out["qp%d_mem" % case], out["qp%d_g" % case], out["qp%d_margin" % case] = np_(mem), np_(g), np.array(margin)
        GEM.project2cone2(g, mem, margin)
        out["qp%d_x" % case] = np_(g)
    save("G6_gem", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9"]
    for w in which:
        globals()[w]()
