"""Shared by the G29 generator (reference run, dev container) and the CPU test of the build's HAT trainer loop
(clsurvey_amd/methods/hat_main.HatTrainer.train): a scripted model (an epoch counter that survives load_state_dict), the
scenarios, and the routine that runs ONE trainer factory (passed in) over them.  The trainer's train_epoch / eval are
replaced by table look-ups on both sides; what is compared is the loop around them: learning rate and lambda per epoch,
warm-up end, patience, stop / suspend, checkpoints, the model kept."""
import os
import shutil
import tempfile
from types import SimpleNamespace

import torch
import torch.nn as nn

_plateau = [0.2, 0.4] + [0.4] * 200
_rising = [min(1.0, 0.02 * (i + 1)) for i in range(200)]
_mixed = ([0.3, 0.2, 0.35, 0.2, 0.2, 0.2, 0.2, 0.2, 0.2, 0.2, 0.2, 0.2, 0.4] + [0.2] * 20 + [0.45] + [0.2] * 200)
SCENARIOS = []
for joint in (True, False):
    for t in (0, 1):
        for warm in ((True, False) if joint else (True,)):
            SCENARIOS += [
                dict(joint=joint, t=t, enable_warmup=warm, val=_plateau, nepochs=100, save_freq=5, resume_after=None),
                dict(joint=joint, t=t, enable_warmup=warm, val=_mixed, nepochs=60, save_freq=4, resume_after=None),
                dict(joint=joint, t=t, enable_warmup=warm, val=_rising, nepochs=14, save_freq=5, resume_after=None),
                dict(joint=joint, t=t, enable_warmup=warm, val=_mixed, nepochs=100, save_freq=5, resume_after=17),
            ]


class ScriptedHat(nn.Module):
    def __init__(self, enable_warmup):
        super().__init__()
        self.w = nn.Parameter(torch.zeros(3))
        self.register_buffer("tick", torch.zeros((), dtype=torch.long))
        self.enable_warmup = enable_warmup

    # what the reference trainer calls on its net around the loop (vgg_hat.py summaries): nothing to do here
    def backmask_summary(self, *a, **k):
        pass

    def premask_summary(self, *a, **k):
        pass


def scripted_train_epoch(trainer, log):
    def train_epoch(t, loader, *a, **k):
        trainer.model.tick += 1
        log.append({"epoch": int(trainer.model.tick), "lr": trainer.optimizer.param_groups[0]["lr"], "lamb": trainer.lamb})
        return 1.0, 0.5
    return train_epoch


def scripted_eval(trainer, val):
    def evaluate(t, loader, *a, **k):
        return 1.0, val[int(trainer.model.tick) - 1]
    return evaluate


def _files(exp_dir):
    out = {}
    for f in sorted(os.listdir(exp_dir)):
        c = torch.load(os.path.join(exp_dir, f), weights_only=False)
        if f == "epoch.pth.tar":
            out[f] = {k: (float(c[k]) if isinstance(c[k], float) else c[k]) for k in ("e", "patience", "best_acc", "lr", "warmup", "smax", "post_lamb")
                      if k in c}
            out[f]["model_tick"] = int(c["model"]["tick"])
            out[f]["optimizer_lr"] = c["optimizer"]["param_groups"][0]["lr"]
        elif f == "best_model.pth.tar":
            out[f] = {"model_tick": int(c.tick), "smax": getattr(c, "smax", None), "lamb": getattr(c, "lamb", None)}
    return out


def run_once(make_trainer, sc, exp_dir, nepochs):
    model = ScriptedHat(sc["enable_warmup"])
    args = SimpleNamespace(save_freq=sc["save_freq"], weight_decay=0.0, parameter=[400.0, 0.75], batch_size=8, lr=0.05, nepochs=nepochs)
    trainer = make_trainer(sc["joint"], model, exp_dir, nepochs, args)
    log = []
    trainer.train_epoch = scripted_train_epoch(trainer, log)
    trainer.eval = scripted_eval(trainer, sc["val"])
    loaders = {"train": [0] * 4, "val": [0] * 2}
    best_model, best_acc = trainer.train(sc["t"], loaders)
    return {"best_acc": float(best_acc), "best_model_tick": int(best_model.tick), "epochs": log, "files": _files(exp_dir),
            "final_optimizer_lr": trainer.optimizer.param_groups[0]["lr"]}


def generate(make_trainer):
    out = []
    for sc in SCENARIOS:
        exp_dir = tempfile.mkdtemp()
        entry = {"scenario": {k: v for k, v in sc.items() if k != "val"}}
        if sc["resume_after"] is None:
            entry["run"] = run_once(make_trainer, sc, exp_dir, sc["nepochs"])
        else:
            entry["first"] = run_once(make_trainer, sc, exp_dir, sc["resume_after"])
            try:
                entry["run"] = run_once(make_trainer, sc, exp_dir, sc["nepochs"])
            except KeyError as e:            # a trainer that cannot read its own checkpoint back: the type is the record
                entry["run"] = {"raises": "KeyError", "key": str(e)}
        out.append(entry)
        shutil.rmtree(exp_dir)
    return out
