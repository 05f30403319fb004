"""Method plugin surface of the Continual Hyperparameter Framework, table-driven.

The framework drivers talk to a method only through `manager.method.<hook>(args, manager, ...)` and a handful of class
attributes (reference: src/methods/method.py — `Method` ABC :81-111, `parse` :35-78, `set_hyperparams` :238-274; probed
hooks: lr_grid_train.py:43,83,157, framework_train.py:82,95,105,276,282, eval.py:46,125,232, inference.py:60).  The build
keeps that surface — names, attributes, return conventions — but not the reference's one-hand-written-class-per-method
layout: a method here is a row of `SPECS` naming

  * its hyper-parameters / static hyper-parameters / framework flags,
  * how phase 1 (LR grid node) and phase 2 (one stability-decay attempt) are run: an entry point of the build's trainers
    plus an ARGUMENT MAP  keyword -> where the value comes from  ("M.x" manager attribute, "A.x" args attribute, "H.x"
    hyper-parameter, "S.x" static hyper-parameter, "lr" the grid LR, anything else a literal),
  * how a batch is evaluated (`get_output`) and how a saved model is tested (`inference_eval`),
  * optional hooks, attached only where the spec lists them — the drivers probe them with hasattr().

`_build_class` turns a row into a class with the reference's name, so `methods.method.EWC`, `parse("EWC")`, pickles
and isinstance checks keep working.
"""
import copy
import itertools
import os
import shutil
import time
import warnings
from collections import OrderedDict
from enum import Enum, auto

import torch

from ..data import DeviceLoader, TensorTaskDataset, load_task_datasets
from . import ebll as _ebll
from . import ewc as _ewc
from . import finetune as _ft
from . import gem_main as _gem
from . import hat_main as _hat
from . import imm as _imm
from . import lwf as _lwf
from . import mas as _mas
from . import packnet_main as _packnet
from . import si as _si
from . import train_common as tc


class Category(Enum):
    MODEL_BASED = auto()
    DATA_BASED = auto()
    MASK_BASED = auto()
    BASELINE = auto()
    REHEARSAL_BASED = auto()


# ------------------------------------------------------------------------------------------------ argument maps
def _dev(args):
    return getattr(args, "device", "cuda")


def _resolve(src, method, args, manager, hp, lr):
    """One argument-map entry -> value."""
    if not isinstance(src, str) or len(src) < 2 or src[1] != ".":
        return lr if src == "lr" else src
    scope, key = src[0], src[2:]
    if scope == "M":
        return getattr(manager, key)
    if scope == "A":
        return getattr(args, key)
    if scope == "a":                                   # optional args attribute, '' when absent
        return getattr(args, key, "")
    if scope == "H":
        return hp[key]
    if scope == "S":
        return method.static_hyperparams[key]
    raise KeyError(src)


def _call(entry, argmap, method, args, manager, hp=None, lr=None, **extra):
    kw = {k: _resolve(v, method, args, manager, hp, lr) for k, v in argmap.items()}
    kw.update(extra)
    return entry(device=_dev(args), **kw)


_TASK_IO = dict(dataset_path="M.current_task_dataset_path", batch_size="A.batch_size", num_epochs="A.num_epochs",
                weight_decay="A.weight_decay", saving_freq="A.saving_freq")

# phase 2: one training of the current task at the hyper-parameters of this decay attempt
PHASE2 = {
    "EWC": (_ewc.fine_tune_EWC_acuumelation, dict(
        _TASK_IO, previous_task_model_path="M.previous_task_model_path", exp_dir="M.heuristic_exp_dir",
        data_dir="A.data_dir", reg_sets="M.reg_sets", reg_lambda="H.lambda", lr="A.lr")),
    "MAS": (_mas.fine_tune_objective_based_acuumelation, dict(
        _TASK_IO, previous_task_model_path="M.previous_task_model_path", init_model_path="A.init_model_path",
        exp_dir="M.heuristic_exp_dir", data_dir="A.data_dir", reg_sets="M.reg_sets", reg_lambda="H.lambda", lr="A.lr",
        norm="L2", b1=False)),
    "SI": (_si.fine_tune_elastic, dict(
        _TASK_IO, model_path="M.previous_task_model_path", exp_dir="M.heuristic_exp_dir", reg_lambda="H.lambda",
        lr="A.lr", init_freeze=0)),
    "LWF": (_lwf.fine_tune_SGD_LwF, dict(
        _TASK_IO, previous_task_model_path="M.previous_task_model_path", init_model_path="a.init_model_path",
        exp_dir="M.heuristic_exp_dir", lr="A.lr", init_freeze=0, last_layer_name="A.classifier_heads_starting_idx",
        reg_lambda="H.lambda")),
    "EBLL": (_ebll.fine_tune_SGD_EBLL, dict(
        _TASK_IO, previous_task_model_path="M.previous_task_model_path",
        autoencoder_model_path="M.autoencoder_model_path", init_model_path="a.init_model_path",
        exp_dir="M.heuristic_exp_dir", lr="A.lr", init_freeze=0, reg_alpha="H.ebll_reg_alpha",
        reg_lambda="H.reg_lambda")),
}

# phase 1 entry points that are plain trainer calls (the shared SGD finetune has its own function below)
PHASE1 = {
    "l2transfer": (_imm.fine_tune_l2transfer, dict(
        _TASK_IO, model_path="M.previous_task_model_path", exp_dir="M.gridsearch_exp_dir", reg_lambda="H.lambda", lr="lr")),
}

# `overwrite_args` dictionaries of the mask / rehearsal trainers (their mains take one dict, like the reference's)
_HAT_ARGS = dict(weight_decay="A.weight_decay", task_name="A.task_name", task_count="A.task_counter",
                 model_name="A.model_name", nepochs="A.num_epochs", cuda=True, dataset_path="M.current_task_dataset_path",
                 dataset="M.dataset", batch_size="A.batch_size", lr="A.lr", approach="hat", save_freq="A.saving_freq")
_GEM_ARGS = dict(weight_decay="A.weight_decay", task_name="A.task_name", task_count="A.task_counter", method="gem",
                 n_memories="S.mem_per_task", n_epochs="A.num_epochs", cuda=True,
                 dataset_path="M.current_task_dataset_path", batch_size="A.batch_size", lr="A.lr")
_PACKNET_ARGS = dict(weight_decay="A.weight_decay", train_path="M.current_task_dataset_path",
                     test_path="M.current_task_dataset_path", finetune_epochs="A.num_epochs", cuda=True,
                     train_bn="A.train_bn", saving_freq="A.saving_freq", current_dataset_idx="A.task_counter")


def _classes_per_task(manager):
    return [len(v) for v in manager.dataset.classes_per_task.values()]


# ------------------------------------------------------------------------------------------------ evaluation strategies
def get_output_def(model, heads, images, current_head_idx, final_layer_idx):
    """Multi-head nets that keep ONE head in the module tree: put the task's head in, forward in eval mode."""
    model.classifier._modules[final_layer_idx] = heads[current_head_idx]
    model.eval()
    with torch.no_grad():
        return model(images)


def _out_swap_head(method, images, args):
    return get_output_def(args.model, args.heads, images, args.current_head_idx, args.final_layer_idx)


def _out_own_heads(method, images, args):
    """LwF / EBLL wrappers return every head's output (EBLL: (heads, codes)); the shared first-task model plain logits."""
    with torch.no_grad():
        out = args.model(images)
    if isinstance(out, tuple):
        out = out[0]
    return out[args.current_head_idx] if isinstance(out, list) else out


def _out_gem_slice(method, images, args):
    lo, hi = args.model.compute_offsets(args.current_head_idx, args.model.cum_nc_per_task)
    return args.model(images, args.current_head_idx)[:, lo:hi]


def _out_hat_gated(method, images, args):
    from . import hat as H
    head = args.heads[args.current_head_idx]
    eng = getattr(args, "_hat_engine", None)
    if eng is None or eng.net is not args.model or eng.net.classifier[0] is not head:
        args.model.classifier = torch.nn.ModuleList([head])
        eng = args._hat_engine = H.HatEngine(args.model, args.batch_size, tuple(images.shape[1:]), images.device)
    return eng.forward(args.task_idx, images, args.model.smax)


OUTPUT = {"swap_head": _out_swap_head, "own_heads": _out_own_heads, "gem_slice": _out_gem_slice, "hat_gated": _out_hat_gated}


def _test(manager, model, args, head_idx, heads):
    from ..framework import inference
    return inference.test_model(manager.method, model, args.dset_path, head_idx, subset=args.test_set, target_head=heads,
                                batch_size=args.batch_size, task_idx=args.eval_dset_idx, device=_dev(args))


def _eval_swap_head(args, manager):
    """Saved model + the one saved head of the evaluated task (Finetune.inference_eval, method.py:1081-1103)."""
    from ..framework import inference
    model = tc.load_model(args.eval_model_path)
    if isinstance(model, dict):
        model = model["model"]
    last = str(len(model.classifier._modules) - 1)
    assert isinstance(model.classifier._modules[last], torch.nn.Linear), "NO VALID HEAD IDX"
    heads = inference.get_prev_heads(args.head_paths, last, _dev(args))
    assert len(heads) == 1
    return _test(manager, model, args, 0, heads)


def _eval_as_is(args, manager):
    """The saved wrapper knows its heads; evaluate the task's index (method.py:1172-1182)."""
    return _test(manager, tc.load_model(args.eval_model_path), args, args.eval_dset_idx, None)


def _eval_wrapper_after_first(args, manager):
    """LwF / EBLL: task 1's model is the shared single-head SI model, later ones are wrappers."""
    return (_eval_as_is if args.trained_model_idx > 0 else _eval_swap_head)(args, manager)


def _eval_packnet(args, manager):
    task_name = manager.dataset.get_taskname(args.eval_dset_idx + 1)
    return _packnet.main(dict(train_path=args.dset_path, test_path=args.dset_path, mode="eval",
                              dataset=_packnet_dataset_name(task_name), loadname=args.eval_model_path, cuda=True,
                              batch_size=args.batch_size, current_dataset_idx=args.eval_dset_idx + 1), device=_dev(args))


EVALUATE = {"swap_head": _eval_swap_head, "as_is": _eval_as_is, "wrapper_after_first": _eval_wrapper_after_first,
            "packnet": _eval_packnet}


# ------------------------------------------------------------------------------------------------ shared phase 1 (SGD)
class ConcatTasks(TensorTaskDataset):
    """Several tasks as one dataset, labels of task j shifted by the class counts of the tasks before it
    (data/imgfolder.py ConcatDatasetDynamicLabels)."""

    def __init__(self, dsets, classes_len):
        shift = [0] + list(itertools.accumulate(classes_len))[:-1]
        super().__init__(torch.cat([d.x for d in dsets]), torch.cat([d.y + s for d, s in zip(dsets, shift)]),
                         [c for d in dsets for c in d.classes])


def compose_dataset(dataset_path, batch_size, device="cuda"):
    """(loaders, sizes, classes) over one or more task files, on the device (method.py:1057-1079 builds the same triple
    around DataLoader(num_workers=4))."""
    splits = ("train", "val")
    tasks = [load_task_datasets(p, device) for p in dataset_path]
    per = {s: [t[s] for t in tasks] for s in splits}
    loaders = {s: DeviceLoader(per[s][0] if len(tasks) == 1 else ConcatTasks(per[s], [len(d.classes) for d in per[s]]),
                               batch_size, True, device) for s in splits}
    return loaders, {s: sum(len(d) for d in per[s]) for s in splits}, {s: [d.classes for d in per[s]] for s in splits}


def _phase1_sgd(method, args, manager, lr):
    """One LR-grid node of the maximal-plasticity search: plain SGD finetune from the previous task's model."""
    paths = manager.current_task_dataset_path
    loaders, sizes, classes = compose_dataset(paths if isinstance(paths, list) else [paths], args.batch_size, _dev(args))
    return _ft.fine_tune_SGD(loaders, sizes, classes, model_path=manager.previous_task_model_path,
                             exp_dir=manager.gridsearch_exp_dir, num_epochs=args.num_epochs, lr=lr,
                             weight_decay=args.weight_decay, enable_resume=True, save_models_mode=True,
                             replace_last_classifier_layer=True, freq=args.saving_freq, device=_dev(args),
                             batch_size=args.batch_size)


def _adopt_grid_winner(args, manager):
    """Methods without a phase 2: the winning grid node IS the task's model; TASK_TRAINING links to it."""
    manager.previous_task_model_path = os.path.join(manager.best_exp_grid_node_dirname, "best_model.pth.tar")
    link = os.path.join(manager.parent_exp_dir, "task_" + str(args.task_counter), "TASK_TRAINING")
    if os.path.islink(link) or os.path.exists(link):
        os.unlink(link)
    os.symlink(os.path.relpath(manager.best_exp_grid_node_dirname, os.path.dirname(link)), link)


# ------------------------------------------------------------------------------------------------ PackNet
def _packnet_dataset_name(task_name):
    return "survey_TASK_" + task_name


def _packnet_init(self):
    self.pruned_savename = None
    self.grid_batch_size = 200          # phase 1 always runs at 200 (method.py:524); tests shrink it


def _packnet_train_init(self, args, manager):
    self.pruned_savename = os.path.join(manager.heuristic_exp_dir, "best_model_PRUNED")


def _packnet_grid_prestep(self, args, manager):
    manager.dataset_name = _packnet_dataset_name(args.task_name)
    manager.disable_pruning_mask = args.task_counter == 1          # task 1 trains every weight (method.py:505)
    if args.task_counter != 1:
        return
    wrapped = os.path.join(manager.ft_parent_exp_dir, manager.base_model.name + "_INIT_WRAPPED.pth")
    if not os.path.exists(wrapped):
        arch = "alexnet" if "alexnet" in manager.base_model.name.lower() else "VGGslim_nopretrain"
        _packnet.main(dict(arch=arch, init_dump=True, cuda=True, loadname=manager.previous_task_model_path,
                           save_prefix=wrapped, last_layer_idx=manager.base_model.last_layer_idx,
                           current_dataset_idx=args.task_counter))
    manager.previous_task_model_path = wrapped


def _packnet_grid_train(self, args, manager, lr):
    kw = {k: _resolve(v, self, args, manager, None, lr) for k, v in _PACKNET_ARGS.items()}
    kw.update(mode="finetune", disable_pruning_mask=manager.disable_pruning_mask, dataset=manager.dataset_name,
              num_outputs=len(manager.dataset.classes_per_task[args.task_name]), loadname=manager.previous_task_model_path,
              lr=lr, save_prefix=os.path.join(manager.gridsearch_exp_dir, "best_model"), batch_size=self.grid_batch_size)
    return None, _packnet.main(kw, device=_dev(args))


def _packnet_grid_poststep(self, args, manager):
    manager.best_finetuned_model_path = os.path.join(manager.best_exp_grid_node_dirname, "best_model.pth.tar")


def _packnet_train(self, args, manager, hyperparams):
    kw = {k: _resolve(v, self, args, manager, hyperparams, None) for k, v in _PACKNET_ARGS.items()}
    kw.update(mode="prune", dataset=_packnet_dataset_name(args.task_name), loadname=manager.best_finetuned_model_path,
              post_prune_epochs=10, prune_perc_per_layer=hyperparams["prune_perc_per_layer"],
              lr=args.lr * 0.1,                                    # post-prune retraining at a tenth of the LR (method.py:437)
              save_prefix=self.pruned_savename, batch_size=args.batch_size)
    manager.overwrite_args = kw
    return None, _packnet.main(kw, device=_dev(args))


def _packnet_init_next_task(self, manager):
    assert self.pruned_savename is not None
    for suffix in ("_final.pth.tar", "_postprune.pth.tar"):
        if os.path.exists(self.pruned_savename + suffix):
            if suffix != "_final.pth.tar":
                warnings.warn("Final file not found(no final file saved if finetune gives no improvement)! Using postprune")
            manager.previous_task_model_path = self.pruned_savename + suffix
            return
    raise Exception("Previous task pruned model final/postprune non-existing: {}".format(self.pruned_savename))


def _packnet_train_args_overwrite(args):
    args.train_bn = "BN" in args.model_name
    print("TRAINING BN PARAMS = ", str(args.train_bn))


# ------------------------------------------------------------------------------------------------ HAT / GEM entry
def _hat_run(self, args, manager, parameter, out_dir, finetune):
    kw = {k: _resolve(v, self, args, manager, None, None) for k, v in _HAT_ARGS.items()}
    kw.update(prev_model_path=manager.previous_task_model_path, output=out_dir, parameter=parameter,
              n_tasks=manager.dataset.task_count, is_scratch_model=args.task_counter == 1,
              nc_per_task=_classes_per_task(manager), finetune_mode=finetune)
    manager.overwrite_args = kw
    return _hat.main(kw, device=_dev(args))


def _hat_grid_train(self, args, manager, lr):
    args.lr = lr
    return _hat_run(self, args, manager, list(self.hyperparams.values()), manager.gridsearch_exp_dir, True)


def _hat_train(self, args, manager, hyperparams):
    return _hat_run(self, args, manager, list(hyperparams.values()), manager.heuristic_exp_dir, False)


def _gem_run(self, args, manager, strength, out_dir, prev=None, finetune=False, postprocess=False):
    nc = _classes_per_task(manager)
    kw = {k: _resolve(v, self, args, manager, None, None) for k, v in _GEM_ARGS.items()}
    kw.update(prev_model_path=manager.previous_task_model_path if prev is None else prev, save_path=out_dir,
              n_outputs=sum(nc), memory_strength=strength, n_tasks=manager.dataset.task_count, finetune=finetune,
              is_scratch_model=args.task_counter == 1, postprocess=postprocess)
    manager.overwrite_args = kw
    return _gem.main(kw, nc, device=_dev(args))


def _gem_grid_train(self, args, manager, lr):
    args.lr = lr
    return _gem_run(self, args, manager, 0, manager.gridsearch_exp_dir, finetune=True)


def _gem_train(self, args, manager, hyperparams):
    return _gem_run(self, args, manager, hyperparams["margin"], manager.heuristic_exp_dir)


def _gem_poststep(self, args, manager):
    """Task 1 is the shared SI model: wrap it with its exemplars once (method.py:298-317); later tasks need nothing."""
    if args.task_counter > 1:
        return
    t0 = time.time()
    target = manager.best_model_path
    if not os.path.exists(target):
        args.lr = getattr(args, "lr", None) or 0.0           # the wrapper's optimizer is rebuilt at the next task
        _gem_run(self, args, manager, self.hyperparams["margin"], target, prev=manager.previous_task_model_path,
                 postprocess=True)
    args.postprocess_time = time.time() - t0
    manager.best_model_path = target


# ------------------------------------------------------------------------------------------------ IMM / LwF / EBLL extras
def _imm_init(self, mode="mode"):
    self.set_mode(mode)


def _imm_set_mode(self, mode):
    if mode not in self.modes:
        raise Exception("NO EXISTING IMM MODE: '{}'".format(mode))
    self.mode = mode
    self.eval_name = self.name + "_" + mode


def _imm_eval_model_preprocessing(self, args):
    return _imm.preprocess_merge_IMM(self, args.models_path, args.datasets_path, args.batch_size, overwrite=True,
                                     device=_dev(args))


def _lwf_init(self, warmup_step=False):
    self.warmup_step = warmup_step


def _lwf_train(self, args, manager, hyperparams):
    """method.py:952-975.  With warmup_step the new head is first trained alone for half the epochs into
    task_<t>/HEAD_TRAINING (the distillation training that follows is started with init_freeze=0, i.e. with a fresh
    head either way — the reference never reads the warmed head back, and neither does this)."""
    if self.warmup_step:
        warm_dir = os.path.join(manager.parent_exp_dir, "task_" + str(args.task_counter), "HEAD_TRAINING")
        _lwf.fine_tune_freeze(dataset_path=manager.current_task_dataset_path, model_path=manager.previous_task_model_path,
                              exp_dir=warm_dir, batch_size=args.batch_size, num_epochs=int(args.num_epochs / 2),
                              lr=args.lr, device=_dev(args))
        args.init_model_path = warm_dir
    entry, argmap = PHASE2["LWF"]
    return _call(entry, argmap, self, args, manager, hp=hyperparams)


def _ebll_prestep(self, args, manager):
    print("AUTOENCODER PHASE: for prev task ", args.task_counter - 1)
    manager.autoencoder_model_path = _ebll_autoencoder_grid(self, args, manager)
    print("AUTOENCODER PHASE DONE")


def _ebll_autoencoder_grid(self, args, manager):
    """(code size, alpha, lr) grid of under-complete autoencoders on the PREVIOUS task's features, scored by the previous
    model's accuracy on the reconstructions (method.py:842-908); finished nodes are checkpointed, losers deleted."""
    S = self.static_hyperparams
    parent = os.path.join(manager.parent_exp_dir, "task_" + str(args.task_counter - 1), "ENCODER_TRAINING")
    ledger_path = os.path.join(parent, "grid_checkpoint.pth")
    ledger = torch.load(ledger_path, weights_only=False) if os.path.exists(ledger_path) else {"header": ("dim", "alpha", "lr")}
    best_dir, best_acc = None, 0
    for node in itertools.product(S["encoder_dims"], S["encoder_alphas"], S["autoencoder_lr"]):
        dim, alpha, lr = node
        node_dir = os.path.join(parent, "dim={}_alpha={}_lr={}".format(dim, alpha, lr))
        if node not in ledger:
            os.makedirs(node_dir, exist_ok=True)
            t0 = time.time()
            _, ledger[node] = _ebll.fine_tune_Adam_Autoencoder(
                dataset_path=args.previous_task_dataset_path, previous_task_model_path=manager.previous_task_model_path,
                exp_dir=node_dir, batch_size=args.batch_size, num_epochs=S["autoencoder_epochs"], lr=lr, alpha=alpha,
                last_layer_name=args.classifier_heads_starting_idx, auto_dim=dim, device=_dev(args))
            args.presteps_elapsed_time += time.time() - t0
            torch.save(ledger, ledger_path)
        acc = ledger[node]
        print("autoencoder acc={}".format(acc))
        loser = node_dir
        if acc > best_acc:
            loser, best_dir, best_acc = best_dir, node_dir, acc
        if loser is not None:
            shutil.rmtree(loser, ignore_errors=True)
    if best_acc < 0.40:
        print("[WARNING] Auto-encoder grid not sufficient: max attainable acc = {}".format(best_acc))
    return os.path.join(best_dir, "best_model.pth.tar")


# ------------------------------------------------------------------------------------------------ the table
def _spec(name, category, hyper=(), static=(), flags=(), phase1="sgd", phase2=None, output="swap_head",
          evaluate="swap_head", init=None, hooks=None, attrs=None, doc=""):
    return dict(name=name, category=category, hyper=tuple(hyper), static=tuple(static), flags=tuple(flags), phase1=phase1,
                phase2=phase2, output=output, evaluate=evaluate, init=init, hooks=hooks or {}, attrs=attrs or {}, doc=doc)


SPECS = [
    _spec("finetuning", Category.BASELINE, flags=("grid_chkpt", "start_scratch", "no_framework"),
          hooks={"grid_poststep": staticmethod(_adopt_grid_winner), "compose_dataset": staticmethod(compose_dataset)},
          doc="plain SGD per task, no forgetting-related mechanism (method.py:994); grid only"),
    _spec("EWC", Category.MODEL_BASED, hyper=[("lambda", 400)], phase2="EWC", doc="method.py:663"),
    _spec("MAS", Category.MODEL_BASED, hyper=[("lambda", 3)], phase2="MAS", doc="method.py:726"),
    _spec("SI", Category.MODEL_BASED, hyper=[("lambda", 400)], phase2="SI", doc="method.py:695"),
    _spec("IMM", Category.MODEL_BASED, hyper=[("lambda", 0.01)], flags=("grid_chkpt", "no_framework"), phase1="l2transfer",
          init=_imm_init, attrs={"modes": ["mean", "mode"]},
          hooks={"set_mode": _imm_set_mode, "grid_poststep": staticmethod(_adopt_grid_winner),
                 "eval_model_preprocessing": _imm_eval_model_preprocessing},
          doc="L2-transfer training per task, mean / mode merge of the task models before evaluation (method.py:760-819)"),
    _spec("LWF", Category.DATA_BASED, hyper=[("lambda", 10)], phase2="LWF", output="own_heads",
          evaluate="wrapper_after_first", init=_lwf_init, hooks={"train": _lwf_train},
          doc="new head per task, old heads distilled from the previous model (method.py:940-989)"),
    _spec("EBLL", Category.DATA_BASED, hyper=[("reg_lambda", 10), ("ebll_reg_alpha", 1)],
          static=[("autoencoder_lr", [0.01]), ("autoencoder_epochs", 50), ("encoder_alphas", [1e-1, 1e-2]),
                  ("encoder_dims", [100, 300])],
          phase2="EBLL", output="own_heads", evaluate="wrapper_after_first",
          hooks={"prestep": _ebll_prestep, "_autoencoder_grid": _ebll_autoencoder_grid},
          doc="LwF + a code loss through one autoencoder per finished task (method.py:822-936)"),
    _spec("packnet", Category.MASK_BASED, hyper=[("prune_perc_per_layer", 0.9)], flags=("grid_chkpt", "start_scratch"),
          phase1=None, evaluate="packnet", init=_packnet_init,
          hooks={"get_dataset_name": staticmethod(_packnet_dataset_name), "train_init": _packnet_train_init,
                 "grid_prestep": _packnet_grid_prestep, "grid_train": _packnet_grid_train,
                 "grid_poststep": _packnet_grid_poststep, "train": _packnet_train,
                 "init_next_task": _packnet_init_next_task,
                 "train_args_overwrite": staticmethod(_packnet_train_args_overwrite)},
          doc="phase 1 = finetune on the free weights per LR, phase 2 = prune + post-prune finetune; one wrapped model with "
              "a head per task and uint8 ownership masks (method.py:415-556)"),
    _spec("HAT", Category.MASK_BASED, hyper=[("smax", 800), ("c", 2.5)], flags=("start_scratch",), phase1=None,
          output="hat_gated", hooks={"grid_train": _hat_grid_train, "train": _hat_train},
          doc="hard attention to the task; accuracies are fractions in [0, 1] (method.py:600-660)"),
    _spec("GEM", Category.REHEARSAL_BASED, hyper=[("margin", 1)], static=[("mem_per_task", 1024)],
          flags=("wrap_first_task_model",), phase1=None, output="gem_slice", evaluate="as_is",
          hooks={"grid_train": _gem_grid_train, "train": _gem_train, "poststep": _gem_poststep},
          doc="gradient episodic memory; task 1 only wraps the shared SI model with its exemplars (method.py:281-412)"),
]


class Method:
    """What the drivers rely on; every concrete method is generated from a SPECS row by _build_class."""
    name = eval_name = None
    category = None
    extra_hyperparams_count = 0
    hyperparams = OrderedDict()
    spec = None

    def __init__(self, *a, **kw):
        self.hyperparams = copy.deepcopy(type(self).hyperparams)
        if type(self).__dict__.get("static_hyperparams") is not None:
            self.static_hyperparams = copy.deepcopy(type(self).static_hyperparams)
        if self.spec["init"] is not None:
            self.spec["init"](self, *a, **kw)

    def grid_train(self, args, manager, lr):
        p1 = self.spec["phase1"]
        if p1 == "sgd":
            return _phase1_sgd(self, args, manager, lr)
        entry, argmap = PHASE1[p1]
        return _call(entry, argmap, self, args, manager, hp=self.hyperparams, lr=lr)

    def train(self, args, manager, hyperparams):
        entry, argmap = PHASE2[self.spec["phase2"]]
        return _call(entry, argmap, self, args, manager, hp=hyperparams)

    def get_output(self, images, args):
        return OUTPUT[self.spec["output"]](self, images, args)

    def inference_eval(self, args, manager):
        return EVALUATE[self.spec["evaluate"]](args, manager)


def _build_class(spec):
    ns = dict(name=spec["name"], eval_name=spec["name"], category=spec["category"], spec=spec, __doc__=spec["doc"],
              extra_hyperparams_count=len(spec["hyper"]), hyperparams=OrderedDict(spec["hyper"]))
    if spec["static"]:
        ns["static_hyperparams"] = OrderedDict(spec["static"])
    ns.update({flag: True for flag in spec["flags"]})
    ns.update(spec["attrs"])
    ns.update(spec["hooks"])
    if spec["phase2"] is None and "train" not in spec["hooks"]:
        ns["train"] = None                      # grid-only methods have no phase 2; the drivers never ask for it
    cls_name = {"finetuning": "Finetune", "packnet": "PackNet"}.get(spec["name"], spec["name"])
    return type(cls_name, (Method,), ns)


_REGISTRY = OrderedDict()
for _s in SPECS:
    _c = _build_class(_s)
    _REGISTRY[_s["name"]] = _c
    globals()[_c.__name__] = _c           # methods.method.EWC, .PackNet, ... as in the reference


def set_hyperparams(method, hyperparams, static_params=False):
    """Command-line override of the (static) hyper-parameters, values in the order of the method's dict.  Grammar of
    method.py:238-274:  '0.5,300' -> first = 0.5, second = 300;  '0.1,0.2;5.2,300' -> first = [0.1, 0.2], second =
    [5.2, 300] (';' separates entries once any of them is a list).  'def' or an empty slot keeps that entry's default;
    a single bare value ('400') is taken as the first entry (the reference stops with a TypeError on it)."""
    assert isinstance(hyperparams, str)
    target = getattr(method, "static_hyperparams", None) if static_params else method.hyperparams
    if target is None:
        return
    keep = object()

    def number(tok):
        tok = tok.strip()
        return keep if tok in ("def", "") else float(tok)

    fields = [f for f in hyperparams.split(";") if f.strip() != ""]
    if len(fields) == 1:
        values = [number(t) for t in fields[0].split(",")]
    else:
        values = []
        for f in fields:
            nums = [v for v in map(number, f.split(",")) if v is not keep]
            values.append(keep if not nums else nums[0] if len(nums) == 1 else nums)
    for key, v in zip(list(target), values):
        if v is not keep:
            target[key] = v
    method.init_hyperparams = copy.deepcopy(target)


def parse(method_name):
    """Method instance for a --method_name (method.py:35-78); 'IMM' names carry the merge mode: modeIMM, IMM_mean, ..."""
    if "IMM" in method_name:
        return _REGISTRY["IMM"](method_name.replace("_", "").replace("IMM", "").strip())
    if method_name in _REGISTRY:
        return _REGISTRY[method_name]()
    raise NotImplementedError("Method not yet parseable: %r" % method_name)


def register(cls):
    _REGISTRY[cls.name] = cls
    return cls
