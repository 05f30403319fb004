"""Experiment (CPU, plain torch): does a 2-task EWC step on the 'blobs' data exercise the stability-decay loop?
Task 1 from torchvision's init, Fisher diagonal as EWC/main_EWC.py:138-157 (batch-summed gradient squared / N), phase 1
(plain finetune of task 2 -> A_ft), phase 2 at lambda = 400, 200, ... until acc >= A_ft (1 - 0.2); reports task-1 accuracy
under the new trunk + old head (forgetting).

  python tools/experiments/synth_ewc_probe.py hw n_train epochs g amp noise_lr noise_px q
"""
import copy
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, __file__.rsplit("/", 1)[0])
import synth_probe as SP   # noqa: E402


def fit(m, xtr, ytr, xva, yva, epochs, lr, omega=None, star=None, lam=0.0, bs=200):
    opt = torch.optim.SGD(m.parameters(), lr=lr, momentum=0.9)
    best, best_state, count = 0.0, None, 0
    for ep in range(epochs):
        if count > 10:
            break
        if count == 5:
            for g in opt.param_groups:
                g["lr"] *= 0.1
        perm = torch.randperm(len(xtr))
        m.train()
        for i in range(0, len(xtr), bs):
            idx = perm[i:i + bs]
            opt.zero_grad()
            F.cross_entropy(m(xtr[idx]), ytr[idx]).backward()
            if omega is not None:
                with torch.no_grad():
                    for p, o, s in zip(list(m.parameters())[:-2], omega, star):
                        p.grad += 2 * lam * o * (p - s)
            opt.step()
        acc = evaluate(m, xva, yva)
        if acc > best:
            best, best_state, count = acc, copy.deepcopy(m.state_dict()), 0
        else:
            count += 1
    m.load_state_dict(best_state)
    return best, ep + 1


def evaluate(m, x, y):
    m.eval()
    with torch.no_grad():
        return sum(int((m(x[i:i + 500]).argmax(1) == y[i:i + 500]).sum()) for i in range(0, len(x), 500)) / len(x)


def main():
    a = sys.argv[1:]
    hw, ntr, epochs, g = int(a[0]), int(a[1]), int(a[2]), int(a[3])
    amp, nlr, npx, q = float(a[4]), float(a[5]), float(a[6]), float(a[7])
    kaiming = len(a) > 8 and a[8] == "k"
    SP.Q[0] = q
    tasks = []
    for t in range(2):
        gen = torch.Generator().manual_seed(7001 + t)
        xtr, ytr, protos = SP.blobs(ntr, 20, hw, g, amp, nlr, npx, gen)
        xva, yva, _ = SP.blobs(ntr // 4, 20, hw, g, amp, nlr, npx, gen, protos)
        tasks.append((xtr, ytr, xva, yva))
    torch.manual_seed(0)
    m1 = SP.make_net(hw, 20, kaiming)
    acc1, ep1 = fit(m1, *tasks[0], epochs, 1e-2)
    print("task 1: val %.3f after %d epochs" % (acc1, ep1), flush=True)
    # Fisher diagonal of task 1
    params = list(m1.parameters())
    omega = [torch.zeros_like(p) for p in params]
    xtr, ytr = tasks[0][0], tasks[0][1]
    m1.eval()
    for i in range(0, ntr, 200):
        m1.zero_grad()
        F.nll_loss(F.log_softmax(m1(xtr[i:i + 200]), 1), ytr[i:i + 200], reduction="sum").backward()
        for o, p in zip(omega, params):
            o += p.grad ** 2 / ntr
    print("omega: sum %.4g max %.4g; max per tensor %s" % (sum(float(o.sum()) for o in omega), max(float(o.max()) for o in omega),
                                                       ["%.2g" % float(o.max()) for o in omega]), flush=True)
    star = [p.detach().clone() for p in params[:-2]]
    omega = omega[:-2]

    def task2_model():
        m = copy.deepcopy(m1)
        m[2][4] = torch.nn.Linear(128, 20)
        return m
    best_ft = 0.0
    for lr in (1e-2, 5e-3, 1e-3):
        torch.manual_seed(1)
        m = task2_model()
        acc, ep = fit(m, *tasks[1], epochs, lr)
        print("phase 1 lr %g: val %.3f (%d epochs)" % (lr, acc, ep), flush=True)
        if acc > best_ft:
            best_ft, best_lr = acc, lr
    lam = 400.0
    for attempt in range(8):
        torch.manual_seed(1)
        m = task2_model()
        acc, ep = fit(m, *tasks[1], epochs, best_lr, omega, star, lam)
        old = copy.deepcopy(m)
        old[2][4] = m1[2][4]
        print("phase 2 lambda %g: val %.3f (%d epochs), threshold %.3f; task 1 under the new trunk: %.3f (was %.3f)"
              % (lam, acc, ep, best_ft * 0.8, evaluate(old, tasks[0][2], tasks[0][3]), acc1), flush=True)
        if acc >= best_ft * 0.8:
            break
        lam *= 0.5


if __name__ == "__main__":
    main()
