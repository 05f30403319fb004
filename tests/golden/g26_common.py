"""Shared by the G26 generator (reference run, dev container) and the CPU test of the build's per-task orchestration
(framework_train.py:219-292): scenarios, a stand-in method that logs every hook the framework calls together with the
fields it can see at that moment, and the routine that runs ONE (framework_single_task, Manager) pair over them."""
import collections
import os
import shutil
import tempfile
from types import SimpleNamespace

import torch

LRS = [1e-2, 1e-3]
HOOKSETS = {"plain": (), "post_next": ("poststep", "init_next_task"), "grid_hooks": ("grid_prestep", "grid_poststep"),
            "phase2_hooks": ("train_init", "prestep", "poststep")}
SCENARIOS = []
for hooks in HOOKSETS:
    for task, first_scratch, wrap, name, keep_ft in ((2, False, False, "standin", False), (3, False, False, "packnet", False),
                                                     (2, False, False, "standin", True), (1, False, False, "standin", False),
                                                     (1, True, False, "standin", False), (1, False, True, "standin", False)):
        SCENARIOS.append(dict(hooks=hooks, task=task, train_first_task=first_scratch, wrap_first_task_model=wrap, name=name,
                              save_models_FT_heuristic=keep_ft))


class _DS:
    name = "standin_ds"

    def get_taskname(self, i):
        return "task%d" % i

    def get_task_dataset_path(self, task_name=None, rnd_transform=False):
        return "data/%s%s.pth" % (task_name, "_rnd" if rnd_transform else "")


class LogMethod:
    eval_name = "standin"

    def __init__(self, name, log):
        self.name = name
        self.hyperparams = collections.OrderedDict([("lambda", 8.0)])
        self.log = log

    def _see(self, what, args, manager, **extra):
        rel = lambda p: os.path.relpath(p, manager.parent_exp_dir) if isinstance(p, str) and p.startswith(manager.parent_exp_dir) else p   # noqa: E731
        row = {"hook": what, "save_models_mode": getattr(args, "save_models_mode", None) if args is not None else None,
               "lr": getattr(args, "lr", None) if args is not None else None,
               "reg_sets": getattr(manager, "reg_sets", None),
               "heads_idx": getattr(args, "classifier_heads_starting_idx", None) if args is not None else None,
               "gridsearch_exp_dir": rel(getattr(manager, "gridsearch_exp_dir", None)),
               "heuristic_exp_dir": rel(getattr(manager, "heuristic_exp_dir", None)),
               "best_model_path": rel(getattr(manager, "best_model_path", None)),
               "previous_task_model_path": rel(manager.previous_task_model_path)}
        row.update(extra)
        self.log.append(row)

    def grid_train(self, args, manager, lr):
        self._see("grid_train", args, manager, grid_lr=lr)
        return None, {1e-2: 0.4, 1e-3: 0.6}[lr]

    def train(self, args, manager, hp):
        self._see("train", args, manager, hp=[float(v) for v in hp.values()])
        os.makedirs(manager.heuristic_exp_dir, exist_ok=True)
        torch.save({"hp": [float(v) for v in hp.values()]}, os.path.join(manager.heuristic_exp_dir, "best_model.pth.tar"))
        return None, 0.9 / (1.0 + 0.2 * float(hp["lambda"]))          # accepted at lambda = 2 against 0.6 * 0.8


def _args_hook(h):
    def fn(self, args, manager):
        self._see(h, args, manager)
    fn.__name__ = fn.__qualname__ = h
    return fn


def _init_next_task(self, manager):
    self._see("init_next_task", None, manager)
    manager.previous_task_model_path = "set_by_init_next_task"


# one importable class per hook set (the reference pickles vars(manager), the method object included, into its checkpoint)
METHODS = {}
for _tag, _hooks in HOOKSETS.items():
    _ns = {h: (_init_next_task if h == "init_next_task" else _args_hook(h)) for h in _hooks}
    METHODS[_tag] = type("LogMethod_" + _tag, (LogMethod,), _ns)
    METHODS[_tag].__module__ = __name__
    globals()["LogMethod_" + _tag] = METHODS[_tag]


def generate(framework_single_task, Manager):
    out = []
    for sc in SCENARIOS:
        root = tempfile.mkdtemp()
        log = []
        meth = METHODS[sc["hooks"]](sc["name"], log)
        base = SimpleNamespace(last_layer_idx=6, name="base", path="models/base.pth")
        mgr = Manager(_DS(), meth, "prev_model.pth", root, base)
        args = SimpleNamespace(task_counter=sc["task"], task_name="task%d" % sc["task"], train_first_task=sc["train_first_task"],
                               wrap_first_task_model=sc["wrap_first_task_model"], save_models_FT_heuristic=sc["save_models_FT_heuristic"],
                               lrs=list(LRS), finetune_iterations=1, max_attempts_per_task=5, inv_drop_margin=0.8, decaying_factor=0.5)
        try:
            framework_single_task(args, mgr)
            ended = "returned"
        except AttributeError as e:
            ended = "AttributeError"
        rel = lambda p: os.path.relpath(p, root) if isinstance(p, str) and p.startswith(root) else p       # noqa: E731
        out.append({"scenario": sc, "ended": ended, "log": log, "previous_task_model_path": rel(mgr.previous_task_model_path),
                    "save_models_mode": getattr(args, "save_models_mode", None), "lr": getattr(args, "lr", None),
                    "previous_task_dataset_path": getattr(args, "previous_task_dataset_path", None),
                    "timers": [getattr(args, k, None) is not None for k in ("phase1_elapsed_time", "presteps_elapsed_time",
                                                                            "convergence_iteration_elapsed_time", "postprocess_time")]})
        shutil.rmtree(root)
    return out
