"""Shared by the G25 generator (reference run, dev container) and the CPU test of the build's evaluation driver and naming
helpers: the cases, the stand-in method, and the routines that run ONE implementation (passed in) over them."""
import collections
import os
import shutil
import tempfile
from types import SimpleNamespace

import torch

NAME_CASES = [
    dict(drop_margin=0.2, decaying_factor=0.5, num_epochs=70, batch_size=200, weight_decay=0, lr_grid=[1e-2, 5e-3, 1e-3, 5e-4, 1e-4],
         model_name="small_VGG9_cl_128_128", hyperparams=[("lambda", 400)], static=None),
    dict(drop_margin=0.2, decaying_factor=0.5, num_epochs=70, batch_size=200, weight_decay=1e-4, lr_grid=[1e-3, 1e-2],
         model_name="base_VGG9_cl_512_512_DROP", hyperparams=[("smax", 800), ("c", 2.5)], static=None),
    dict(drop_margin=0.5, decaying_factor=0.25, num_epochs=1, batch_size=20, weight_decay=0.0, lr_grid=[0.01],
         model_name="deep_VGG22_cl_512_512_BN", hyperparams=[("reg_lambda", 10), ("ebll_reg_alpha", 1)],
         static=[("autoencoder_lr", [0.01]), ("autoencoder_epochs", 50), ("encoder_alphas", [0.1, 0.01]), ("encoder_dims", [100, 300])]),
    dict(drop_margin=0.2, decaying_factor=0.5, num_epochs=50, batch_size=128, weight_decay=5e-4, lr_grid=[5e-4, 1e-4, 1e-2],
         model_name="wide_VGG9_cl_512_512_DROP_BN", hyperparams=[("prune_perc_per_layer", 0.9)], static=None),
    dict(drop_margin=0.2, decaying_factor=0.5, num_epochs=70, batch_size=200, weight_decay=0, lr_grid=[1e-2],
         model_name="alexnet_pretrained", hyperparams=[("margin", 1.0)], static=[("mem_per_task", 1024)]),
    dict(drop_margin=0.2, decaying_factor=0.5, num_epochs=70, batch_size=200, weight_decay=0, lr_grid=[1e-2, 5e-3],
         model_name="small_VGG9_cl_128_128", hyperparams=[], static=None),
]


def names(get_exp_name, get_init_modelname):
    out = []
    for c in NAME_CASES:
        args = SimpleNamespace(**{k: c[k] for k in ("drop_margin", "decaying_factor", "num_epochs", "batch_size", "weight_decay",
                                                    "lr_grid", "model_name")})
        meth = SimpleNamespace(hyperparams=collections.OrderedDict(c["hyperparams"]))
        if c["static"] is not None:
            meth.static_hyperparams = collections.OrderedDict(c["static"])
        out.append({"exp_name": get_exp_name(args, meth), "first_task_modelname": get_init_modelname(args)})
    return out


# ---------------------------------------------------------------------------------------------- evaluation driver
T = 4
ACC = [[90.0 - 11.0 * d - 3.5 * (m - d) - 0.25 * d * m for m in range(T)] for d in range(T)]       # acc[task][model]
EVAL_CASES = [
    dict(tag="all", start=1, stop=4, fail=None, overwrite=False, debug=False, preexisting=None),
    dict(tag="window", start=2, stop=3, fail=None, overwrite=False, debug=False, preexisting=None),
    dict(tag="fails_later_model", start=1, stop=4, fail=[1, 2], overwrite=False, debug=False, preexisting=None),
    dict(tag="fails_first_model", start=1, stop=4, fail=[2, 2], overwrite=False, debug=False, preexisting=None),
    dict(tag="already_done", start=1, stop=4, fail=None, overwrite=False, debug=False, preexisting=1),
    dict(tag="overwrite", start=1, stop=4, fail=None, overwrite=True, debug=False, preexisting=1),
    dict(tag="debug", start=1, stop=4, fail=None, overwrite=False, debug=True, preexisting=1),
]


class EvalMethod:
    name = eval_name = "standin"

    def __init__(self, fail):
        self.calls, self.fail = [], fail

    def inference_eval(self, args, manager):
        self.calls.append([args.eval_dset_idx, args.trained_model_idx, args.dset_path, args.head_paths, args.eval_model_path])
        if self.fail is not None and [args.eval_dset_idx, args.trained_model_idx] == self.fail:
            raise RuntimeError("evaluation of this model fails")
        return ACC[args.eval_dset_idx][args.trained_model_idx]


def _plain(v):
    if isinstance(v, dict):
        return [[str(k), _plain(x)] for k, x in v.items()]
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    return v


def evals(eval_all_models_all_tasks, perf_filename):
    out = []
    for c in EVAL_CASES:
        root = tempfile.mkdtemp()
        meth = EvalMethod(c["fail"])
        mgr = SimpleNamespace(method=meth)
        args = SimpleNamespace(test_starting_task_count=c["start"], test_max_task_count=c["stop"], out_path=root,
                               test_overwrite_mode=c["overwrite"], debug=c["debug"])
        if c["preexisting"] is not None:
            torch.save({"old": True}, os.path.join(root, perf_filename(meth.eval_name, c["preexisting"])))
        ds_paths = ["ds%d" % i for i in range(T)]
        model_paths = ["model%d" % i for i in range(T)]
        eval_all_models_all_tasks(args, mgr, ds_paths, model_paths)
        files = {}
        for f in sorted(os.listdir(root)):
            files[f] = _plain(torch.load(os.path.join(root, f), weights_only=False))
        out.append({"tag": c["tag"], "calls": meth.calls, "files": files})
        shutil.rmtree(root)
    return out


# ---------------------------------------------------------------------------------------------- adopting the grid winner
def adopt(grid_poststep):
    """Methods without a phase 2 (finetuning, IMM): grid_poststep(args, manager) makes the winning grid node the task's
    model and links TASK_TRAINING to it (method.py:1028-1041).  Run twice (a re-run finds the link in place)."""
    root = tempfile.mkdtemp()
    out = []
    try:
        for winner in ("lr=1.0E-03", "lr=5.0E-03_it1"):
            node = os.path.join(root, "task_2", "FT_LR_GRIDSEARCH", winner)
            os.makedirs(node, exist_ok=True)
            mgr = SimpleNamespace(parent_exp_dir=root, best_exp_grid_node_dirname=node, previous_task_model_path="before")
            grid_poststep(SimpleNamespace(task_counter=2), mgr)
            link = os.path.join(root, "task_2", "TASK_TRAINING")
            out.append({"is_link": os.path.islink(link), "target": os.readlink(link),
                        "resolves_to": os.path.relpath(os.path.realpath(link), root),
                        "previous_task_model_path": os.path.relpath(mgr.previous_task_model_path, root)})
    finally:
        shutil.rmtree(root)
    return out
