"""Shared epoch / phase / batch loop of the reference's train_model variants, on NetEngine.

Mirrors
  Finetune/train_SGD.py:41-189   (plain SGD;            70 epochs, stop when count > 10)
  EWC/train_EWC.py:111-234       (optimizer.step(reg_params); same schedule)
  MAS/train_MAS.py:208-335       (same as EWC)
  SI/train_SI.py:152-283         (range(start, num_epochs + 1); stop when count >= 10)
Differences from the reference that do not change results: the per-batch `.item()` syncs
(train_EWC.py:196-197) are replaced by on-device accumulators read once per phase, and batches come
from HBM (clsurvey_amd.data.DeviceLoader) in the same order DataLoader would produce.
"""
import math
import os
import time

import torch

from ..net import NetEngine
from ..data import DeviceLoader


def set_lr(optimizer, lr, count, early_stop="gt"):
    """train_EWC.py:89-101 / train_SGD.py:10-30 ('gt': stop when count > 10);
    SI/train_SI.py:129-141 ('ge': stop when count >= 10). LR x0.1 when count == 5."""
    continue_training = True
    if (count > 10) if early_stop == "gt" else (count >= 10):
        continue_training = False
        print("training terminated")
    if count == 5:
        lr = lr * 0.1
        print("lr is set to {}".format(lr))
        for param_group in optimizer.param_groups:
            param_group["lr"] = lr
    return optimizer, lr, continue_training


class PatiencePlan:
    """Validation-driven schedule of the mask-based trainers (HAT/approaches/hat.py:150-166, hat_finetune.py:108-124):
    a new best resets the patience; otherwise it counts down, the LR is divided by `factor` when half is left and
    training stops at zero.  observe() returns 'best' | 'decay' | 'stop' | 'hold'; `frozen` epochs (HAT's warm-up)
    neither count down nor stop.  stop_at_or_below_zero: joint HAT keeps counting below zero while its first task is
    still under the minimum epoch count, the phase-1 search stops exactly at zero."""

    def __init__(self, lr, patience, factor, stop_at_or_below_zero=True):
        self.lr, self.full, self.factor = lr, patience, factor
        self.patience, self.best = patience, 0
        self.leq = stop_at_or_below_zero

    def restore(self, lr, patience, best):
        self.lr, self.patience, self.best = lr, patience, best

    def observe(self, acc, frozen=False):
        if acc > self.best:
            self.best, self.patience = acc, self.full
            return "best"
        if frozen:
            return "hold"
        self.patience -= 1
        if self.patience == self.full // 2:
            self.lr /= self.factor
            return "decay"
        if (self.patience <= 0) if self.leq else (self.patience == 0):
            return "stop"
        return "hold"


def make_loaders(dsets, batch_size, device, phases=("train", "val"), shuffle=True):
    return {x: DeviceLoader(dsets[x], batch_size, shuffle, device) for x in phases}


def engine_for(model, loaders, batch_size, device):
    any_loader = next(iter(loaders.values()))
    in_shape = tuple(any_loader.x.shape[1:])
    return NetEngine(model, batch_size, in_shape, device)


def save_model(model, path):
    """torch.save(model) as in train_EWC.py:211; the arena marker must not be pickled."""
    rp = getattr(model, "reg_params", None)
    marker = rp.pop("__arena__", None) if isinstance(rp, dict) else None
    try:
        torch.save(model, path)
    finally:
        if marker is not None:
            rp["__arena__"] = marker


def train_model(model, engine, optimizer, lr, dset_loaders, dset_sizes, num_epochs, exp_dir="./", resume="",
                saving_freq=5, step_fn=None, early_stop="gt", extra_epoch=False, save_models_mode=True,
                abort_on_bad_loss=True):
    """Returns (model, best_val_acc in [0,1]).  step_fn() applies the optimizer for one batch
    (default: optimizer.step(model.reg_params))."""
    since = time.time()
    val_beat_counts = 0
    best_acc = 0.0
    start_epoch = 0
    if resume and os.path.isfile(resume):
        checkpoint = torch.load(resume, weights_only=False)
        start_epoch = checkpoint["epoch"]
        best_acc = checkpoint["best_acc"]
        model.load_state_dict(checkpoint["state_dict"])     # in place: parameters AND BatchNorm buffers, by name
        optimizer.load_state_dict(checkpoint["optimizer"])   # momentum goes back into the arena (optim._SGDBase)
        lr = checkpoint["lr"]
        val_beat_counts = checkpoint["val_beat_counts"]
        print("=> loaded checkpoint '{}' (epoch {})".format(resume, checkpoint["epoch"]))
    if step_fn is None:
        def step_fn():
            optimizer.step(model.reg_params)
    stats = torch.zeros(2, dtype=torch.float64, device=engine.device)
    last_epoch = num_epochs + 1 if extra_epoch else num_epochs
    epoch_acc = 0.0
    for epoch in range(start_epoch, last_epoch):
        print("Epoch {}/{}".format(epoch, last_epoch - 1))
        for phase in ("train", "val"):
            if phase == "train":
                optimizer, lr, cont = set_lr(optimizer, lr, val_beat_counts, early_stop)
                if not cont:
                    print("Training complete in {:.0f}s, best val acc {:.4f}".format(time.time() - since, best_acc))
                    return model, best_acc
            model.train(phase == "train")       # train_SGD.py:97-99, train_EWC.py:157-159: Dropout / BatchNorm mode
            stats.zero_()
            for inputs, labels in dset_loaders[phase]:
                engine.loss_step(inputs, labels, "ce_mean", backward=(phase == "train"), stats=stats)
                if phase == "train":
                    step_fn()
            s = stats.cpu()     # the only host sync of the phase
            epoch_loss = float(s[0]) / dset_sizes[phase]
            epoch_acc = float(s[1]) / dset_sizes[phase]
            print("{} Loss: {:.4f} Acc: {:.4f}".format(phase, epoch_loss, epoch_acc))
            if abort_on_bad_loss and (epoch_loss > 1e4 or math.isnan(epoch_loss)):
                return model, best_acc            # train_EWC.py:204-205
            if phase == "val":
                if epoch_acc > best_acc:
                    best_acc = epoch_acc
                    if save_models_mode:
                        save_model(model, os.path.join(exp_dir, "best_model.pth.tar"))
                    val_beat_counts = 0
                else:
                    val_beat_counts += 1
        if save_models_mode and epoch % saving_freq == 0:
            rp = getattr(model, "reg_params", None)
            marker = rp.pop("__arena__", None) if isinstance(rp, dict) else None
            torch.save({"epoch": epoch + 1, "lr": lr, "val_beat_counts": val_beat_counts, "epoch_acc": epoch_acc,
                        "best_acc": best_acc, "arch": "alexnet", "model": model, "state_dict": model.state_dict(),
                        "optimizer": optimizer.state_dict()}, os.path.join(exp_dir, "epoch.pth.tar"))
            if marker is not None:
                rp["__arena__"] = marker
    print("Training complete in {:.0f}s, best val acc {:.4f}".format(time.time() - since, best_acc))
    return model, best_acc


def replace_head(model, n_out):
    """model.classifier[last] = nn.Linear(in_features, n_out) — main_EWC.py:49-53, utils.py:68-72."""
    import torch.nn as nn
    last = str(len(model.classifier._modules) - 1)
    num_ftrs = model.classifier._modules[last].in_features
    model.classifier._modules[last] = nn.Linear(num_ftrs, n_out)
    return model


def load_model(path):
    return torch.load(path, weights_only=False)


def save_preprocessing_time(exp_dir, t):
    os.makedirs(exp_dir, exist_ok=True)
    torch.save(t, os.path.join(exp_dir, "preprocess_time.pth.tar"))    # utils.py:100-105
