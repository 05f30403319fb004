"""Shared by the G23 generator (reference run, dev container) and the CPU test of the build's LR grid: the seeded accuracy
tables, the stand-in method, and the routine that runs ONE grid function over them and records what it decided and left
on disk.  No reference code: the grid function is passed in."""
import os
import shutil
import tempfile
from types import SimpleNamespace

import numpy as np

LRS = [1e-2, 5e-3, 1e-3, 5e-4, 1e-4]                      # framework/main.py:61
MODES = ["all", "only_keep_best", "keep_none"]


def tables(n=36, seed=23):
    gen = np.random.RandomState(seed)
    out = []
    for i in range(n):
        its = 1 + i % 3
        levels = [0.0, 0.05, 0.1, 0.25, 0.5, 0.5, 0.75, 0.9][:3 + i % 6]          # few levels: ties are common
        acc = [[float(levels[gen.randint(len(levels))]) for _ in range(its)] for _ in LRS]
        if i % 12 == 11:
            acc = [[0.0] * its for _ in LRS]                                      # nothing ever beats best_acc = 0
        out.append({"iterations": its, "acc": acc})
    return out


class TableMethod:
    name = eval_name = "table"

    def __init__(self, acc, fail_after=None):
        self.acc, self.calls, self.fail_after = acc, [], fail_after

    def grid_train(self, args, manager, lr):
        if self.fail_after is not None and len(self.calls) >= self.fail_after:
            raise KeyboardInterrupt("interrupted run")
        node = os.path.basename(manager.gridsearch_exp_dir)
        it = int(node.rsplit("_it", 1)[1]) if "_it" in node else 0
        self.calls.append((lr, it))
        with open(os.path.join(manager.gridsearch_exp_dir, "marker.txt"), "w") as f:
            f.write("%r %d" % (lr, it))
        return None, self.acc[LRS.index(lr)][it]


def survivors(ft_dir):
    return sorted(d for d in os.listdir(ft_dir) if d.startswith("lr=") and os.path.isdir(os.path.join(ft_dir, d)))


def run(grid_fn, tab, mode, root, fail_after=None, method=None):
    ds = SimpleNamespace(get_taskname=lambda i: "task%d" % i)
    meth = method or TableMethod(tab["acc"], fail_after)
    mgr = SimpleNamespace(dataset=ds, method=meth, parent_exp_dir=root)
    args = SimpleNamespace(task_counter=2, lrs=list(LRS), finetune_iterations=tab["iterations"])
    best_lr, best_acc = grid_fn(args, mgr, mode)
    ft = mgr.ft_parent_exp_dir
    return {"best_lr": best_lr, "best_acc": best_acc,
            "best_node": os.path.basename(mgr.best_exp_grid_node_dirname) if mgr.best_exp_grid_node_dirname else None,
            "survivors": survivors(ft), "calls": [[lr, it] for lr, it in meth.calls],
            "ft_dir": os.path.relpath(ft, root), "task_name": args.task_name}


def checkpointed(root, rec):
    import torch
    ck = torch.load(os.path.join(root, rec["ft_dir"], "grid_checkpoint.pth"), weights_only=False)
    return [[lr, [float(a) for a in ck["processed_lrs"][lr]["acc"]]] for lr in ck["processed_lrs"]]


def generate(grid_fn):
    out = []
    for i, tab in enumerate(tables()):
        entry = {"iterations": tab["iterations"], "acc": tab["acc"], "modes": {}}
        for mode in MODES:
            root = tempfile.mkdtemp()
            rec = run(grid_fn, tab, mode, root)
            rec["checkpoint"] = checkpointed(root, rec)
            entry["modes"][mode] = rec
            shutil.rmtree(root)
        # interrupted after k nodes, resumed on the same tree
        k = 1 + (i * 7) % (len(LRS) * tab["iterations"] - 1) if len(LRS) * tab["iterations"] > 1 else 1
        root = tempfile.mkdtemp()
        try:
            run(grid_fn, tab, "only_keep_best", root, fail_after=k)
            raise SystemExit("the interrupted run finished?")
        except KeyboardInterrupt:
            pass
        rec = run(grid_fn, tab, "only_keep_best", root)
        rec["checkpoint"] = checkpointed(root, rec)
        entry["resumed"] = {"interrupted_after": k, **rec}
        shutil.rmtree(root)
        out.append(entry)
    return out
