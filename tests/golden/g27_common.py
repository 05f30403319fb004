"""Shared by the G27 generator (reference run, dev container) and the CPU test of the build's epoch loop
(clsurvey_amd/methods/train_common.py): a scripted network whose accuracy per epoch and phase is a table, an optimizer
that logs the learning rate it is stepped with, the scenarios, and the routine that runs ONE implementation of the four
train_model variants (passed in) over them and records what the loop did."""
import os
import shutil
import tempfile

import torch
import torch.nn as nn
import torch.nn.functional as F

C, B, NT, NV = 4, 4, 2, 2                      # classes, batch size, train / val batches per epoch
SIZES = {"train": B * NT, "val": B * NV}
VARIANTS = ["sgd", "ewc", "mas", "si", "lwf", "imm", "ebll"]  # Finetune/train_SGD.py, EWC/train_EWC.py, MAS/train_MAS.py, SI/train_SI.py, LwF/main_LWF.py, IMM/train_L2transfer.py, EBLL/Finetune_SGD_EBLL.py

_plateau = [0.25, 0.5] + [0.5] * 40
_rising = [min(1.0, 0.125 * (i + 1)) for i in range(8)] + [1.0] * 40
_mixed = [0.5, 0.25, 0.625, 0.25, 0.25, 0.625, 0.75, 0.5, 0.5, 0.5, 0.75, 0.5, 0.5, 0.875] + [0.5] * 40
SCENARIOS = [
    dict(tag="plateau", val=_plateau, num_epochs=30, saving_freq=5, nan_at=None, save_models_mode=True, resume_after=None),
    dict(tag="rising_short", val=_rising, num_epochs=9, saving_freq=5, nan_at=None, save_models_mode=True, resume_after=None),
    dict(tag="mixed", val=_mixed, num_epochs=25, saving_freq=2, nan_at=None, save_models_mode=True, resume_after=None),
    dict(tag="nan_loss", val=_mixed, num_epochs=12, saving_freq=5, nan_at=3, save_models_mode=True, resume_after=None),
    dict(tag="resumed", val=_mixed, num_epochs=30, saving_freq=5, nan_at=None, save_models_mode=True, resume_after=7),
    dict(tag="no_saving", val=_plateau, num_epochs=8, saving_freq=5, nan_at=None, save_models_mode=False, resume_after=None),
]


class ScriptedNet(nn.Module):
    """forward() classifies the first round(acc * N) samples of the running phase correctly, acc from the table; the call
    counter is a buffer, so a resumed run (load_state_dict) carries on in the right epoch."""

    def __init__(self, val, nan_at):
        super().__init__()
        self.w = nn.Parameter(torch.zeros(1))
        self.register_buffer("calls", torch.zeros((), dtype=torch.long))
        self.val, self.nan_at = list(val), nan_at
        self.reg_params = {}

    def forward(self, x):
        epoch, r = divmod(int(self.calls), NT + NV)
        phase = "train" if r < NT else "val"
        self.calls += 1
        acc = self.val[epoch] if phase == "val" else 0.5
        label, idx = x[:, 0].long(), x[:, 1].long()
        target = torch.where(idx < round(acc * SIZES[phase]), label, (label + 1) % C)
        logits = 5.0 * F.one_hot(target, C).float() + 0.0 * self.w
        if self.nan_at is not None and epoch == self.nan_at and phase == "train":
            logits = logits * float("nan")
        return logits


class ScriptedLwf(ScriptedNet):
    """the same network behind LwF's interface: a list of logits per head, the new task's head last (main_LWF.py:176-178)"""

    def forward(self, x):
        new = super().forward(x)
        return [0.0 * new.detach() + 0.0 * self.w, new]


class ScriptedEbll(ScriptedNet):
    """EBLL's interface: (logits per head, codes per old task) — Finetune_SGD_EBLL.py:318"""

    def forward(self, x):
        new = super().forward(x)
        return [0.0 * new.detach() + 0.0 * self.w, new], [torch.zeros(x.shape[0], 3) + 0.0 * self.w]


class ScriptedEbllTeacher(nn.Module):
    def forward(self, x):
        return [torch.zeros(x.shape[0], C)], [torch.zeros(x.shape[0], 3)]


class ScriptedTeacher(nn.Module):
    """the frozen previous model: one old head, constant logits"""

    def forward(self, x):
        return [torch.zeros(x.shape[0], C)]


class LogSGD(torch.optim.SGD):
    """SGD that accepts the reg_params argument of the penalised optimizers and logs the LR of every step."""

    def __init__(self, params, lr):
        super().__init__(params, lr=lr, momentum=0.9)
        self.trace = []

    def step(self, reg_params=None, closure=None):
        self.trace.append(self.param_groups[0]["lr"])
        return super().step()


def loaders():
    out = {}
    for phase, nb in (("train", NT), ("val", NV)):
        rows = []
        for b in range(nb):
            idx = torch.arange(b * B, (b + 1) * B)
            labels = idx % C
            rows.append((torch.stack([labels.float(), idx.float()], 1), labels))
        out[phase] = rows
    return out


def _files(exp_dir):
    out = {}
    for f in sorted(os.listdir(exp_dir)):
        if f == "epoch.pth.tar":
            c = torch.load(os.path.join(exp_dir, f), weights_only=False)
            out[f] = {k: (float(c[k]) if k != "epoch" and k != "val_beat_counts" else int(c[k]))
                      for k in ("epoch", "lr", "val_beat_counts", "epoch_acc", "best_acc")}
            out[f]["model_calls"] = int(c["state_dict"]["calls"])
        elif f == "best_model.pth.tar":
            out[f] = {"model_calls": int(torch.load(os.path.join(exp_dir, f), weights_only=False).calls)}
        elif f == "preprocess_time.pth.tar":
            out[f] = {}                                    # (a wall-clock value: presence only)
    return out


def run_once(train, variant, sc, exp_dir, num_epochs, resume):
    model = {"lwf": ScriptedLwf, "ebll": ScriptedEbll}.get(variant, ScriptedNet)(sc["val"], sc["nan_at"])
    opt = LogSGD(model.parameters(), lr=0.01)
    _, best = train(variant, model, opt, 0.01, loaders(), dict(SIZES), num_epochs, exp_dir, resume, sc["saving_freq"],
                    sc["save_models_mode"])
    per_epoch = [opt.trace[i] for i in range(0, len(opt.trace), NT)]
    return {"best_acc": float(best), "forward_calls": int(model.calls), "lr_per_epoch": per_epoch, "final_lr": opt.param_groups[0]["lr"],
            "files": _files(exp_dir)}


def generate(train):
    """train(variant, model, optimizer, lr, loaders, sizes, num_epochs, exp_dir, resume, saving_freq, save_models_mode)
    -> (model, best_acc)"""
    out = []
    for sc in SCENARIOS:
        for variant in VARIANTS:
            if not sc["save_models_mode"] and variant != "sgd":
                continue                                  # only train_SGD.train_model has the switch
            exp_dir = tempfile.mkdtemp()
            entry = {"tag": sc["tag"], "variant": variant}
            if sc["resume_after"] is None:
                entry["run"] = run_once(train, variant, sc, exp_dir, sc["num_epochs"], "")
            else:
                entry["first"] = run_once(train, variant, sc, exp_dir, sc["resume_after"], "")
                entry["run"] = run_once(train, variant, sc, exp_dir, sc["num_epochs"], os.path.join(exp_dir, "epoch.pth.tar"))
            out.append(entry)
            shutil.rmtree(exp_dir)
    return out
