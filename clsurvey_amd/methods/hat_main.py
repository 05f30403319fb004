"""HAT trainer on the HIP path — mirror of src/methods/HAT/run.py:main, approaches/hat.py:Appr (joint
training with annealed gates, sparsity regulariser, warm-up on the first task, patience schedule) and
approaches/hat_finetune.py:Appr (phase-1 maximal-plasticity search: all gates open, back-mask only).

One batch of Appr.train_epoch (hat.py:200-249) = HatEngine.step (gates -> gate-folded weights -> ONE
clhip_net_loss_step -> gate / embedding gradient kernels) + HAT_SGD.step + clhip_clamp on the embeddings;
running loss / accuracy are device counters read once per epoch.
"""
import argparse
import os
import time
from copy import deepcopy

import torch
from ..data import load_task_datasets

from ..data import DeviceLoader
from . import hat as H


def set_lr_(optimizer, lr):
    """HAT_utils.py:72-74."""
    for param_group in optimizer.param_groups:
        param_group["lr"] = lr


def get_model(model):
    """HAT_utils.py:47-48."""
    return deepcopy(model)


class Appr(object):
    """approaches/hat.py:13-299."""

    def __init__(self, model, exp_dir, nepochs=100, sbatch=200, lr=0.05, lr_min=1e-4, lr_factor=3, lr_patience=10,
                 clipgrad=10000, args=None, in_shape=None, device="cuda"):
        self.model = model
        self.exp_dir = exp_dir
        self.save_freq = args.save_freq
        self.momentum = 0.9
        self.weight_decay = args.weight_decay
        self.nepochs = nepochs
        self.sbatch = sbatch
        self.lr = lr
        self.lr_min = lr_min
        self.lr_factor = lr_factor
        self.lr_patience = lr_patience
        self.clipgrad = clipgrad
        self.device = torch.device(device)
        self.hat = H.HatEngine(model, sbatch, in_shape, device)
        self.optimizer = self._get_optimizer()
        assert len(args.parameter) == 2
        self.smax = args.parameter[0]
        self.post_lamb = args.parameter[1]
        self.warmup_lamb = 0
        self.lamb = None
        self.warmup_lr = 0.01
        self.enable_warmup = self.model.enable_warmup
        self.warmup_epochs = 10
        self.min_epochs = int(self.nepochs / 2)
        self.mask_pre, self.mask_back = None, {}
        self._stats = torch.zeros(2, dtype=torch.float64, device=self.device)
        self._reg = torch.zeros((), dtype=torch.float64, device=self.device)
        print("smax={},post_lamb={}, enable_warmup={}, warmup_lamb={}, warmup_epochs={}".format(
            self.smax, self.post_lamb, self.enable_warmup, self.warmup_lamb, self.warmup_epochs))

    def init_masks(self, current_task, smax):
        """hat.py:57-89."""
        return H.init_masks(self.hat, current_task, smax)

    def _get_optimizer(self, lr=None):
        if lr is None:
            lr = self.lr
        return H.HAT_SGD(self.model.parameters(), lr=lr, momentum=self.momentum, weight_decay=self.weight_decay)

    def _rebind(self, model):
        """self.model <- model (a deepcopy or a loaded one): new engine/arena over ITS parameters."""
        self.model = model
        in_shape = self.hat.engine.in_shape
        self.hat = H.HatEngine(model, self.sbatch, in_shape, self.device)

    def train(self, t, dset_loaders, eps=1e-6):
        """hat.py:95-198. Returns (best validation model, best validation accuracy in [0, 1])."""
        loaded_chkpt = False
        chkpt_path = os.path.join(self.exp_dir, "epoch.pth.tar")
        if os.path.exists(chkpt_path):
            chkpt = torch.load(chkpt_path, weights_only=False)
            try:
                assert abs(self.smax - chkpt["smax"]) < eps
                assert abs(self.post_lamb - chkpt["post_lamb"]) < eps
                init_e = chkpt["e"]
                with torch.no_grad():
                    for (_, p), (_, v) in zip(self.model.state_dict().items(), chkpt["model"].items()):
                        p.copy_(v)
                self.optimizer.load_state_dict(chkpt["optimizer"])
                best_acc = deepcopy(chkpt["best_acc"])
                lr = deepcopy(chkpt["lr"])
                patience = deepcopy(chkpt["patience"])
                warmup = deepcopy(chkpt["warmup"])
                loaded_chkpt = True
            except Exception as e:
                print("No chkpt loaded:{}".format(e))
        if not loaded_chkpt:
            patience = self.lr_patience
            best_acc = 0
            init_e = 0
            warmup = t == 0 and self.enable_warmup
            lr = self.lr if not warmup else self.warmup_lr
            self.optimizer = self._get_optimizer(lr)
        best_model = get_model(self.model)
        self.mask_pre, self.mask_back = self.init_masks(t, self.smax)

        for e in range(init_e, self.nepochs):
            self.lamb = self.warmup_lamb if warmup else self.post_lamb
            clock0 = time.time()
            train_loss, train_acc = self.train_epoch(t, dset_loaders["train"])
            clock1 = time.time()
            print("| Epoch {:3d}, time={:5.1f}ms | Train: loss={:.6f}, acc={:5.1f}% |".format(
                e + 1, 1000 * self.sbatch * (clock1 - clock0) / len(dset_loaders["train"]), train_loss,
                100 * train_acc), end="")
            valid_loss, valid_acc = self.eval(t, dset_loaders["val"])
            print(" Valid: loss={:.6f}, acc={:5.1f}% | lamb={:.4f} |".format(valid_loss, 100 * valid_acc, self.lamb),
                  end="")
            if valid_acc > best_acc:
                best_acc = valid_acc
                best_model = get_model(self.model)
                patience = self.lr_patience
                print(" *", end="")
                torch.save(best_model, os.path.join(self.exp_dir, "best_model.pth.tar"))
            elif not warmup:
                patience -= 1
                if patience == self.lr_patience // 2:
                    lr /= self.lr_factor
                    print(" lr={:.1e}".format(lr), end="")
                    set_lr_(self.optimizer, lr)
                elif patience <= 0:
                    if e < self.min_epochs and t == 0:
                        print("[BREAK SUSPEND] need at least {} epochs".format(self.min_epochs), end="")
                    else:
                        print("[BREAK] Patience=0/{}, with lr={:.1e}".format(self.lr_patience, lr))
                        break
            if warmup and e >= self.warmup_epochs:
                warmup = False
                patience = self.lr_patience
                set_lr_(self.optimizer, self.lr)
                print("[WARMUP END] Lambda_pre -> lambda_post (lr={})".format(self.lr), end="")
            if (e + 1) % self.save_freq == 0:
                torch.save({"post_lamb": self.post_lamb, "smax": self.smax, "warmup": warmup, "e": e + 1,
                            "patience": patience, "best_acc": best_acc, "lr": lr,
                            "optimizer": self.optimizer.state_dict(), "model": self.model.state_dict()}, chkpt_path)
                print(" -> chkpt", end="")
            print()

        self.model = best_model
        self.model.smax = self.smax
        self.model.lamb = self.lamb
        torch.save(self.model, os.path.join(self.exp_dir, "best_model.pth.tar"))
        return self.model, best_acc

    def _epoch_stats(self, n):
        s = self._stats.cpu()
        return float(s[0]) / n, float(s[1]) / n

    def train_epoch(self, t, dset_loader, thres_cosh=50, thres_emb=6):
        """hat.py:200-249. loss logged = CE + lamb*reg, like the reference."""
        self._stats.zero_()
        self._reg.zero_()
        total_num = 0
        nb = len(dset_loader)
        batch_idx = 0
        count = None
        for images, targets in dset_loader:
            bs = images.shape[0]
            progress_ratio = batch_idx / (nb - 1)
            batch_idx += 1
            assert 0 <= progress_ratio <= 1
            s = (self.smax - 1 / self.smax) * progress_ratio + 1 / self.smax
            if count is None and self.mask_pre is None:
                count = float(sum(g.numel() for g in self.hat.gate))     # task 0: numel of all gates (hat.py:294)
            elif count is None:
                count = float(sum(float((1 - mp).sum().item()) for mp in self.mask_pre))
            # stats accumulates sum_i CE_i (mean * bs) and hits, on the device
            _, reg, _ = self.hat.step(t, images, targets, s, self.mask_pre, self.lamb, count, backward=True,
                                      stats=self._stats)
            self._reg += reg.double() * bs
            self.optimizer.step(self.model, self.mask_back, t, s, thres_cosh, self.smax, self.clipgrad)
            H.clamp_embeddings(self.model, float(thres_emb))
            total_num += bs
        ce, acc = self._epoch_stats(total_num)
        return ce + float(self._reg.item()) / total_num, acc

    def eval(self, t, dset_loader):
        """hat.py:251-283."""
        self._stats.zero_()
        self._reg.zero_()
        total_num = 0
        for images, targets in dset_loader:
            bs = images.shape[0]
            _, reg, _ = self.hat.step(t, images, targets, self.smax, self.mask_pre, self.lamb, None, backward=False,
                                      stats=self._stats)
            self._reg += reg.double() * bs
            total_num += bs
        ce, acc = self._epoch_stats(total_num)
        reg = float(self._reg.item()) / total_num
        print("<reg={:.6f}/ce={:.6f}>".format(reg, ce), end="")
        return ce + reg, acc


class ApprFinetune(Appr):
    """approaches/hat_finetune.py:13-178: every unit open (mask of ones), CE only, back-mask on the gradients."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.nepochs += self.warmup_epochs

    def _get_optimizer(self, lr=None):
        if lr is None:
            lr = self.lr
        return H.HAT_SGD([p for p in self.model.parameters() if p.requires_grad], lr=lr, momentum=self.momentum,
                         weight_decay=self.weight_decay)

    def train(self, t, dset_loaders):
        self.mask_pre, self.mask_back = self.init_masks(t, self.smax)
        chkpt_path = os.path.join(self.exp_dir, "epoch.pth.tar")
        if os.path.exists(chkpt_path):
            chkpt = torch.load(chkpt_path, weights_only=False)
            init_e = deepcopy(chkpt["e"])
            with torch.no_grad():
                for (_, p), (_, v) in zip(self.model.state_dict().items(), chkpt["model"].items()):
                    p.copy_(v)
            self.optimizer.load_state_dict(chkpt["optimizer"])
            best_acc = deepcopy(chkpt["best_acc"])
            lr = deepcopy(chkpt["lr"])
            patience = deepcopy(chkpt["patience"])
        else:
            patience = self.lr_patience
            best_acc = 0
            init_e = 0
            lr = self.lr
            self.optimizer = self._get_optimizer(lr)
        best_model = get_model(self.model)
        for e in range(init_e, self.nepochs):
            clock0 = time.time()
            train_loss, train_acc = self.train_epoch(t, dset_loaders["train"])
            clock1 = time.time()
            print("| Epoch {:3d}, time={:5.1f}ms | Train: loss={:.3f}, acc={:5.1f}% |".format(
                e + 1, 1000 * self.sbatch * (clock1 - clock0) / len(dset_loaders["train"]), train_loss,
                100 * train_acc), end="")
            valid_loss, valid_acc = self.eval(t, dset_loaders["val"])
            print(" Valid: loss={:.3f}, acc={:5.1f}% |".format(valid_loss, 100 * valid_acc), end="")
            if valid_acc > best_acc:
                best_acc = valid_acc
                best_model = get_model(self.model)
                patience = self.lr_patience
                print(" *", end="")
                if os.path.exists(self.exp_dir):
                    torch.save(best_model, os.path.join(self.exp_dir, "best_model.pth.tar"))
            else:
                patience -= 1
                if patience == self.lr_patience // 2:
                    lr /= self.lr_factor
                    print(" lr={:.1e}".format(lr), end="")
                    set_lr_(self.optimizer, lr)
                elif patience == 0:
                    print("[BREAK] Patience=0/{}, with lr={:.1e}".format(self.lr_patience, lr))
                    break
            if (e + 1) % self.save_freq == 0:
                torch.save({"model": self.model.state_dict(), "e": e + 1, "patience": patience, "best_acc": best_acc,
                            "lr": lr, "optimizer": self.optimizer.state_dict()}, chkpt_path)
                print(" -> chkpt", end="")
            print()
        return best_model, best_acc

    def _pass(self, dset_loader, backward, t=None):
        self._stats.zero_()
        total_num = 0
        for images, targets in dset_loader:
            self.hat.plain_step(images, targets, backward=backward, stats=self._stats)
            if backward:
                self.optimizer.step(self.model, self.mask_back, t, finetune=True)
            total_num += images.shape[0]
        return self._epoch_stats(total_num)

    def train_epoch(self, t, dset_loader, thres_cosh=50, thres_emb=6):
        return self._pass(dset_loader, True, t)

    def eval(self, t, dset_loader):
        return self._pass(dset_loader, False)


def main(overwrite_args, device="cuda"):
    """run.py:9-120."""
    tstart = time.time()
    parser = argparse.ArgumentParser()
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--approach", default="", type=str)
    parser.add_argument("--output", default="", type=str)
    parser.add_argument("--nepochs", default=200, type=int)
    parser.add_argument("--save_freq", default=20, type=int)
    parser.add_argument("--lr", default=1e10, type=float)
    parser.add_argument("--parameter", type=str, default="")
    args = parser.parse_known_args([])[0]
    for key_arg, val_arg in overwrite_args.items():
        setattr(args, key_arg, val_arg)
    args.task_idx = args.task_count - 1
    if args.approach != "hat":
        raise NotImplementedError("Method {} not implemented!".format(args.approach))   # pathnet: out of scope
    if "VGG" not in args.model_name:
        raise NotImplementedError("HAT on the HIP path covers the VGG family (vgg_hat.py), not: " + args.model_name)

    dsets = load_task_datasets(args.dataset_path)
    args.task_imgfolders = dsets
    args.dset_loaders = {x: DeviceLoader(dsets[x], args.batch_size, True, device) for x in ["train", "val"]}
    taskcla = [(t, nc) for t, nc in enumerate(args.nc_per_task)]
    inputsize = (3,) + tuple(args.dataset.input_size)

    if args.is_scratch_model:
        assert args.task_idx == 0
        raw_model = torch.load(args.prev_model_path, weights_only=False)
        net = H.HatNet(raw_model, inputsize, taskcla).to(device)
    else:
        net = torch.load(args.prev_model_path, weights_only=False).to(device)
    cls = ApprFinetune if args.finetune_mode else Appr
    appr = cls(net, args.output, sbatch=args.batch_size, nepochs=args.nepochs, lr=args.lr, args=args, lr_factor=2,
               lr_patience=30, in_shape=inputsize, device=device)
    best_val_model, best_val_acc = appr.train(args.task_idx, args.dset_loaders)
    print("[Elapsed time = {:.1f} h]".format((time.time() - tstart) / (60 * 60)))
    return best_val_model, best_val_acc
