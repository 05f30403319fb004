"""Shared by the G30 generator (reference run, dev container) and the CPU test of the build's PackNet session loops
(clsurvey_amd/methods/packnet_main.Manager.train / prune): a Manager of the given class is created WITHOUT its constructor
(no data, no device), its do_epoch / eval become table look-ups and its pruner a logger; what is recorded is the loop:
learning rate per epoch, early stop, resume fields, files written and what they hold, the order of the prune stage."""
import json
import os
import shutil
import tempfile
from types import SimpleNamespace

import torch
import torch.nn as nn

_plateau = [40.0, 55.0] + [55.0] * 60
_mixed = [50.0, 45.0, 60.0, 45.0, 45.0, 60.0, 65.0, 45.0, 45.0, 45.0, 45.0, 45.0, 70.0] + [45.0] * 60
_falling = [30.0, 25.0, 20.0] + [10.0] * 60
TRAIN_CASES = [
    dict(tag="plateau", val=_plateau, epochs=30, saving_freq=4, start=(0, 0, 0), best_accuracy=0, save=True),
    dict(tag="mixed", val=_mixed, epochs=25, saving_freq=5, start=(0, 0, 0), best_accuracy=0, save=True),
    dict(tag="below_given_best", val=_falling, epochs=20, saving_freq=3, start=(0, 0, 0), best_accuracy=50.0, save=True),
    dict(tag="resumed_fields", val=_mixed, epochs=25, saving_freq=5, start=(6, 3, 60.0), best_accuracy=60.0, save=True),
    dict(tag="no_save", val=_mixed, epochs=9, saving_freq=100, start=(0, 0, 0), best_accuracy=0, save=False),
]
PRUNE_CASES = [dict(tag="prune_then_finetune", post_prune_epochs=12, val=_mixed), dict(tag="prune_only", post_prune_epochs=0, val=_mixed)]


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.shared = nn.Sequential(nn.Linear(2, 2))

    def train_nobn(self, mode=True):
        self.train(mode)


class Pruner:
    current_dataset_idx = 2

    def __init__(self, log):
        self.log, self.current_masks = log, {}

    def prune(self):
        self.log.append("pruner.prune")

    def get_biases(self):
        return {}


def build(Manager, extra, args, val, log):
    m = object.__new__(Manager)
    m.args, m.model, m.pruner = args, Net(), Pruner(log)
    m.dataset2idx, m.dataset2biases = {}, {}
    m.epochs_seen = 0
    for k, v in extra(m).items():
        setattr(m, k, v)

    def do_epoch(epoch_idx, optimizer, set_cuda_hack=False, mem_snapshotted=True):
        m.epochs_seen += 1
        log.append(["epoch", epoch_idx, optimizer.param_groups[0]["lr"], bool(set_cuda_hack)])
        return [50.0]

    def evaluate(dataset_idx, biases=None):
        assert dataset_idx == 2
        log.append("eval")
        return [100.0 - val[max(m.epochs_seen - 1, 0)]]

    m.do_epoch, m.eval = do_epoch, evaluate
    return m


def _args(root, saving_freq, start, post_prune_epochs=0):
    return SimpleNamespace(lr=0.01, saving_freq=saving_freq, starting_epoch=start[0], val_beat_counts=start[1], best_val_acc=start[2],
                           train_bn=False, cuda=False, dataset="survey_TASK_2", train_biases=False, weight_decay=0.0,
                           post_prune_epochs=post_prune_epochs, save_prefix=os.path.join(root, "best_model_PRUNED"))


def _files(root):
    out = {}
    for f in sorted(os.listdir(root)):
        p = os.path.join(root, f)
        if f.endswith(".pth.tar"):
            c = torch.load(p, weights_only=False)
            if not (isinstance(c, dict) and "epoch" in c):
                continue                                   # (the reference's memory-statistics dump shares the suffix)
            out[f] = {k: c[k] for k in ("epoch", "accuracy", "errors", "val_beat_counts", "best_val_acc")}
            out[f]["dataset2idx"] = dict(c["dataset2idx"])
        elif f.endswith(".json"):
            out[f] = {"error_history": json.load(open(p))["error_history"]}
    return out


def generate(Manager, extra, make_optimizer):
    """extra(manager) -> attributes this implementation's Manager needs besides the common ones."""
    out = []
    for c in TRAIN_CASES:
        root = tempfile.mkdtemp()
        log = []
        m = build(Manager, extra, _args(root, c["saving_freq"], c["start"]), c["val"], log)
        opt = make_optimizer(m)
        acc = m.train(c["epochs"], opt, save=c["save"], savename=os.path.join(root, "best_model"), best_accuracy=c["best_accuracy"])
        out.append({"tag": c["tag"], "returned": float(acc), "log": log, "files": _files(root), "final_lr": opt.param_groups[0]["lr"]})
        shutil.rmtree(root)
    for c in PRUNE_CASES:
        root = tempfile.mkdtemp()
        log = []
        m = build(Manager, extra, _args(root, 5, (0, 0, 0), c["post_prune_epochs"]), c["val"], log)
        m.check = lambda verbose=False: log.append("check")
        acc = m.prune()
        out.append({"tag": c["tag"], "returned": None if acc is None else float(acc), "log": log, "files": _files(root)})
        shutil.rmtree(root)
    return out
