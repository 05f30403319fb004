"""Shared by the G24 generator (reference run, dev container) and the CPU test of the build's phase 2: scenarios, the stand-in
method and the routine that runs ONE (HyperparameterFramework, Manager) pair through them and records what it did.  No
reference code: both classes are passed in."""
import collections
import operator
import os
import shutil
import tempfile
from types import SimpleNamespace

import torch

# accuracy model of the stand-in: acc = base / (1 + weight . hyperparams); rises as the hyper-parameters decay
SCENARIOS = []
for hp, op in (([("lambda", 400.0)], None), ([("smax", 800.0), ("c", 2.5)], None), ([("a", 8.0), ("b", 4.0), ("c", 2.0)], None),
               ([("margin", 1.0), ("k", 3.0)], "sub")):
    for ft_acc, margin, max_att, factor in ((0.5, 0.8, 10, 0.5), (0.9, 0.95, 4, 0.5), (0.3, 0.5, 1, 0.1), (0.7, 0.8, 6, 0.25),
                                            (0.05, 0.8, 10, 0.5), (0.99, 1.0, 3, 0.5)):
        SCENARIOS.append(dict(hyperparams=hp, op=op, finetune_acc=ft_acc, inv_drop_margin=margin, max_attempts=max_att,
                              decaying_factor=factor))


class DecayMethod:
    """train() scores the hyper-parameters it is handed, drops a marker with them into the attempt directory (the
    framework removes a failed attempt's directory) and can be told to stop the run after a number of calls."""
    name = eval_name = "standin"

    def __init__(self, scenario, fail_after=None):
        self.hyperparams = collections.OrderedDict(scenario["hyperparams"])
        if scenario["op"] == "sub":
            self.decay_operator = operator.sub
        self.calls, self.fail_after = [], fail_after

    def train(self, args, manager, hp):
        if self.fail_after is not None and len(self.calls) >= self.fail_after:
            raise RuntimeError("interrupted run")
        vals = [float(v) for v in hp.values()]
        self.calls.append(vals)
        os.makedirs(manager.heuristic_exp_dir, exist_ok=True)
        torch.save({"hp": vals, "lr": args.lr}, os.path.join(manager.heuristic_exp_dir, "best_model.pth.tar"))
        return None, 0.6 / (1.0 + 0.01 * sum(abs(v) for v in vals))


class _DS:
    """picklable stand-in of the dataset object (the reference pickles vars(manager) into hyperparams.pth.tar)"""
    name = "standin_ds"

    def get_taskname(self, i):
        return "task%d" % i


def _state(hf):
    return {"hyperparams": [[k, float(v)] for k, v in hf.hyperparams.items()],
            "backup": [[k, float(v)] for k, v in hf.hyperparams_backup.items()],
            "idx": int(hf.hyperparam_idx), "attempts": int(hf.attempts)}


def _snapshot(mgr, hf, meth):
    d = mgr.heuristic_exp_dir
    out = {"calls": meth.calls, "state": _state(hf), "token": os.path.exists(mgr.get_success_token_path(d)),
           "files": sorted(os.listdir(d)) if os.path.isdir(d) else None,
           "best_model_path": os.path.relpath(mgr.best_model_path, mgr.parent_exp_dir) if getattr(mgr, "best_model_path", None) else None}
    model = os.path.join(d, "best_model.pth.tar")
    out["kept_model"] = torch.load(model, weights_only=False) if os.path.exists(model) else None
    ck = os.path.join(d, "hyperparams.pth.tar")
    if os.path.exists(ck):
        c = torch.load(ck, weights_only=False)
        st = c["state"]
        out["checkpoint"] = {"acc_threshold": float(c["acc_threshold"]), "val_acc": float(c["val_acc"]),
                             "state": {"hyperparams": [[k, float(v)] for k, v in st["hyperparams"].items()],
                                       "backup": [[k, float(v)] for k, v in st["hyperparams_backup"].items()],
                                       "idx": int(st["hyperparam_idx"]), "attempts": int(st["attempts"])}}
    return out


def run_one(Framework, Manager, sc, root, fail_after=None):
    meth = DecayMethod(sc, fail_after)
    mgr = Manager(_DS(), meth, "prev_model", root, None)
    args = SimpleNamespace(task_counter=3, max_attempts_per_task=sc["max_attempts"], inv_drop_margin=sc["inv_drop_margin"],
                           decaying_factor=sc["decaying_factor"])
    hf = Framework(meth)
    try:
        hf.stabilityDecay(args, mgr, 5e-3, sc["finetune_acc"])
        ended = "returned"
    except SystemExit as e:                         # the framework's answer to a training that raises (framework_train.py:106-108)
        ended = "exit(%s)" % (e.code,)
    rec = _snapshot(mgr, hf, meth)
    rec["ended"], rec["lr"] = ended, getattr(args, "lr", None)
    return rec


def generate(Framework, Manager):
    out = []
    for i, sc in enumerate(SCENARIOS):
        entry = {"scenario": sc}
        root = tempfile.mkdtemp()
        entry["fresh"] = run_one(Framework, Manager, sc, root)
        entry["again"] = run_one(Framework, Manager, sc, root)             # SUCCESS.FLAG present: phase 2 is skipped
        shutil.rmtree(root)
        n_calls = len(entry["fresh"]["calls"])
        if n_calls > 1:
            k = 1 + i % (n_calls - 1)
            root = tempfile.mkdtemp()
            entry["interrupted"] = dict(after=k, **run_one(Framework, Manager, sc, root, fail_after=k))
            entry["resumed"] = run_one(Framework, Manager, sc, root)       # new objects, state from hyperparams.pth.tar
            shutil.rmtree(root)
        out.append(entry)
    return out
