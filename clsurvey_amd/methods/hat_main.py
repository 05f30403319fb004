"""HAT trainer on the HIP path: what src/methods/HAT/run.py:main drives — approaches/hat.py (joint training: annealed
gates, sparsity regulariser, warm-up on the first task) and approaches/hat_finetune.py (phase-1 maximal-plasticity
search: every unit open, back-mask only) — expressed as ONE epoch driver over two small pieces of policy:

  * `PatiencePlan` (train_common.py): the validation-driven schedule both reference files spell out inline
    (hat.py:150-172, hat_finetune.py:108-124): reset on a new best, LR / lr_factor at half patience, stop at zero;
  * a `_Pass` object that knows how to push one loader through the HatEngine in either mode.

One joint batch (hat.py:200-249) = HatEngine.step (gates -> gate-folded weights -> ONE clhip_net_loss_step -> gate /
embedding gradient kernels) + HAT_SGD.step + clhip_clamp on the embeddings; loss / accuracy are device counters read once
per epoch.  Files written are the reference's: best_model.pth.tar (pickled net) and epoch.pth.tar with the keys of
hat.py:176-181 / hat_finetune.py:127-130, so a run can be resumed by either implementation.
"""
import os
import time
from copy import deepcopy
from types import SimpleNamespace

import torch

from ..data import DeviceLoader, load_task_datasets
from . import hat as H
from .train_common import PatiencePlan

# run.py:107-109 passes these to every approach; hat.py:16-17 holds the rest as defaults
LR_FACTOR, LR_PATIENCE, CLIPGRAD = 2, 30, 10000
WARMUP_LR, WARMUP_EPOCHS, WARMUP_LAMB = 0.01, 10, 0
THRES_COSH, THRES_EMB = 50, 6


class _Pass:
    """One sweep of a loader through the HatEngine.  joint=True: gated forward at annealed s with the sparsity
    regulariser (hat.py:200-283); joint=False: plain forward, CE only (hat_finetune.py:135-178)."""

    def __init__(self, owner, joint):
        self.o, self.joint = owner, joint
        self.stats = torch.zeros(2, dtype=torch.float64, device=owner.device)
        self.reg = torch.zeros((), dtype=torch.float64, device=owner.device)

    def _gate_budget(self):
        """Denominator of the regulariser (hat.py:285-299): free gate capacity left by the earlier tasks."""
        o = self.o
        if o.mask_pre is None:
            return float(sum(g.numel() for g in o.hat.gate))
        return float(sum(float((1 - mp).sum().item()) for mp in o.mask_pre))

    def run(self, t, loader, train):
        o = self.o
        o.model.train(train)             # hat.py:201 / :257 — the plan executor reads the mode (Dropout masks) from the
        o.hat.view.train(train)          # module tree it was built over, the HatNet's plain view
        self.stats.zero_()
        self.reg.zero_()
        seen, nb = 0, len(loader)
        budget = self._gate_budget() if (self.joint and train) else None
        for i, (images, targets) in enumerate(loader):
            bs = images.shape[0]
            if not self.joint:
                o.hat.plain_step(images, targets, backward=train, stats=self.stats)
                if train:
                    o.optimizer.step(o.model, o.mask_back, t, finetune=True)
            else:
                # gate temperature: linear in the batch index over the epoch while training, smax for evaluation
                s = (o.smax - 1 / o.smax) * (i / (nb - 1)) + 1 / o.smax if train else o.smax
                _, reg, _ = o.hat.step(t, images, targets, s, o.mask_pre, o.lamb, budget, backward=train, stats=self.stats)
                self.reg += reg.double() * bs
                if train:
                    o.optimizer.step(o.model, o.mask_back, t, s, THRES_COSH, o.smax, CLIPGRAD, thres_emb=float(THRES_EMB))
            seen += bs
        s = self.stats.cpu()
        ce, acc = float(s[0]) / seen, float(s[1]) / seen
        reg = float(self.reg.item()) / seen if self.joint else 0.0
        return ce, reg, acc


class HatTrainer:
    """Trains task t of a HatNet and returns (best validation model, best validation accuracy in [0, 1])."""

    def __init__(self, model, exp_dir, args, in_shape, joint, device="cuda"):
        self.model, self.exp_dir, self.joint = model, exp_dir, joint
        self.device = torch.device(device)
        self.sbatch, self.base_lr = args.batch_size, args.lr
        self.save_freq, self.weight_decay = args.save_freq, args.weight_decay
        assert len(args.parameter) == 2
        self.smax, self.post_lamb = args.parameter
        self.lamb = None
        # the phase-1 search runs the warm-up's epochs on top (hat_finetune.py:48)
        self.nepochs = args.nepochs + (0 if joint else WARMUP_EPOCHS)
        self.hat = H.HatEngine(model, self.sbatch, in_shape, device)
        self.optimizer = None
        self.mask_pre, self.mask_back = None, {}
        self._pass = _Pass(self, joint)
        print("smax={},post_lamb={}, enable_warmup={}, warmup_lamb={}, warmup_epochs={}".format(
            self.smax, self.post_lamb, model.enable_warmup, WARMUP_LAMB, WARMUP_EPOCHS))

    # kept under the reference's names: tests and the Method class reach for them
    def init_masks(self, current_task, smax):
        return H.init_masks(self.hat, current_task, smax)

    def eval(self, t, loader):
        ce, reg, acc = self._pass.run(t, loader, train=False)
        return ce + reg, acc

    def train_epoch(self, t, loader):
        ce, reg, acc = self._pass.run(t, loader, train=True)
        return ce + reg, acc

    def _new_optimizer(self, lr):
        params = self.model.parameters() if self.joint else [p for p in self.model.parameters() if p.requires_grad]
        return H.HAT_SGD(params, lr=lr, momentum=0.9, weight_decay=self.weight_decay)

    def _resume(self, path, plan):
        """epoch.pth.tar of an interrupted run of the SAME (smax, c); returns (first epoch, warm-up flag) or None."""
        if not os.path.exists(path):
            return None
        ck = torch.load(path, weights_only=False)
        if self.joint and (abs(self.smax - ck.get("smax", 1e30)) > 1e-6 or abs(self.post_lamb - ck.get("post_lamb", 1e30)) > 1e-6):
            print("No chkpt loaded: other hyper-parameters")
            return None
        self.model.load_state_dict(ck["model"])
        self.optimizer = self._new_optimizer(ck["lr"])
        self.optimizer.load_state_dict(ck["optimizer"])
        plan.restore(ck["lr"], ck["patience"], ck["best_acc"])
        return ck["e"], bool(ck.get("warmup", False))

    def train(self, t, loaders):
        plan = PatiencePlan(self.base_lr, LR_PATIENCE, LR_FACTOR, stop_at_or_below_zero=self.joint)
        ck_path = os.path.join(self.exp_dir, "epoch.pth.tar")
        resumed = self._resume(ck_path, plan)
        if resumed is not None:
            first_epoch, warmup = resumed
        else:
            first_epoch = 0
            warmup = self.joint and t == 0 and self.model.enable_warmup      # hat.py:121
            if warmup:
                plan.lr = WARMUP_LR
            self.optimizer = self._new_optimizer(plan.lr)
        best_model = deepcopy(self.model)
        self.mask_pre, self.mask_back = self.init_masks(t, self.smax)
        min_epochs = int(self.nepochs / 2)                                   # first task only (hat.py:163-165)

        for e in range(first_epoch, self.nepochs):
            if self.joint:                                                   # (the phase-1 search has no regulariser: hat_finetune.py:68-71)
                self.lamb = WARMUP_LAMB if warmup else self.post_lamb
            t0 = time.time()
            tr_loss, tr_acc = self.train_epoch(t, loaders["train"])
            ms = 1000 * self.sbatch * (time.time() - t0) / len(loaders["train"])
            va_loss, va_acc = self.eval(t, loaders["val"])
            line = "| Epoch {:3d}, time={:5.1f}ms | Train: loss={:.6f}, acc={:5.1f}% | Valid: loss={:.6f}, acc={:5.1f}% |".format(
                e + 1, ms, tr_loss, 100 * tr_acc, va_loss, 100 * va_acc)
            verdict = plan.observe(va_acc, frozen=warmup)
            if verdict == "best":
                best_model = deepcopy(self.model)
                line += " *"
                if self.joint or os.path.exists(self.exp_dir):
                    torch.save(best_model, os.path.join(self.exp_dir, "best_model.pth.tar"))
            elif verdict == "decay":
                line += " lr={:.1e}".format(plan.lr)
                for g in self.optimizer.param_groups:
                    g["lr"] = plan.lr
            elif verdict == "stop":
                if self.joint and t == 0 and e < min_epochs:
                    line += " [BREAK SUSPEND] need at least {} epochs".format(min_epochs)
                else:
                    print(line + " [BREAK] Patience=0/{}, with lr={:.1e}".format(LR_PATIENCE, plan.lr))
                    break
            if warmup and e >= WARMUP_EPOCHS:                                 # hat.py:167-172
                # the optimizer goes to the task LR, but the schedule keeps counting from the warm-up LR: the reference
                # updates only the optimizer here, so a later decay sets warmup_lr / lr_factor (kept, G12 pins the trace)
                warmup = False
                plan.patience = LR_PATIENCE
                for g in self.optimizer.param_groups:
                    g["lr"] = self.base_lr
                line += " [WARMUP END] lambda -> {} (lr={})".format(self.post_lamb, self.base_lr)
            if (e + 1) % self.save_freq == 0:
                state = {"model": self.model.state_dict(), "e": e + 1, "patience": plan.patience, "best_acc": plan.best,
                         "lr": plan.lr, "optimizer": self.optimizer.state_dict()}
                if self.joint:
                    state.update(post_lamb=self.post_lamb, smax=self.smax, warmup=warmup)
                torch.save(state, ck_path)
                line += " -> chkpt"
            print(line)

        if self.joint:                # hat.py:185-190: the best model carries its gate temperature and lambda
            best_model.smax, best_model.lamb = self.smax, self.lamb
            torch.save(best_model, os.path.join(self.exp_dir, "best_model.pth.tar"))
            self.model = best_model
        return best_model, plan.best


# the Method class and older tests construct these by the reference's names
def Appr(model, exp_dir, nepochs=100, sbatch=200, lr=0.05, args=None, in_shape=None, device="cuda", **_):
    a = SimpleNamespace(**vars(args))
    a.nepochs, a.batch_size, a.lr = nepochs, sbatch, lr
    return HatTrainer(model, exp_dir, a, in_shape, joint=True, device=device)


def ApprFinetune(model, exp_dir, nepochs=100, sbatch=200, lr=0.05, args=None, in_shape=None, device="cuda", **_):
    a = SimpleNamespace(**vars(args))
    a.nepochs, a.batch_size, a.lr = nepochs, sbatch, lr
    return HatTrainer(model, exp_dir, a, in_shape, joint=False, device=device)


_DEFAULTS = dict(seed=0, approach="", output="", nepochs=200, save_freq=20, lr=1e10, parameter="")     # run.py:13-22


def main(overwrite_args, device="cuda"):
    """run.py:9-120: `overwrite_args` is the dict methods/method.py hands over (keys at method.py:638-658)."""
    tstart = time.time()
    args = SimpleNamespace(**{**_DEFAULTS, **overwrite_args})
    args.task_idx = args.task_count - 1
    if args.approach != "hat":
        raise NotImplementedError("Method {} not implemented!".format(args.approach))   # pathnet: out of scope
    if "VGG" in args.model_name:
        make_net = H.HatNet                          # run.py:59-60
    elif "alexnet" in args.model_name:
        make_net = H.HatNetAlexnet                   # run.py:62-63
    else:
        raise NotImplementedError("HAT on the HIP path covers the VGG family and AlexNet (vgg_hat.py, alexnet_hat.py), not: "
                                  + args.model_name)

    dsets = load_task_datasets(args.dataset_path)
    args.task_imgfolders = dsets
    args.dset_loaders = {x: DeviceLoader(dsets[x], args.batch_size, True, device) for x in ("train", "val")}
    taskcla = list(enumerate(args.nc_per_task))
    inputsize = (3,) + tuple(args.dataset.input_size)

    prev = torch.load(args.prev_model_path, weights_only=False)
    if args.is_scratch_model:
        assert args.task_idx == 0
        net = make_net(prev, inputsize, taskcla).to(device)          # first task: wrap the raw model (run.py:84-92)
    else:
        net = prev.to(device)
    trainer = HatTrainer(net, args.output, args, inputsize, joint=not args.finetune_mode, device=device)
    best_val_model, best_val_acc = trainer.train(args.task_idx, args.dset_loaders)
    print("[Elapsed time = {:.1f} h]".format((time.time() - tstart) / 3600))
    return best_val_model, best_val_acc
