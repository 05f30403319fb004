"""Method plugin surface — mirror of src/methods/method.py (Method ABC :81-111, parse :35,
set_hyperparams :238-274, classes EWC :663, SI :695, MAS :726, Finetune :994).

Same class attributes (name, eval_name, category, extra_hyperparams_count, hyperparams) and hooks
(grid_train, train, inference_eval, get_output, grid_poststep, compose_dataset) as the reference,
so the framework drivers only talk to `manager.method.<hook>(args, manager, ...)`.
"""
import copy
import os
import time
import warnings
from abc import ABC, abstractmethod
from collections import OrderedDict
from enum import Enum, auto

import torch

from ..data import DeviceLoader, TensorTaskDataset, load_task_datasets
from . import ebll as trainEBLL
from . import ewc as trainEWC
from . import finetune as trainFT
from . import gem_main as trainRehearsal
from . import hat_main as trainHAT
from . import imm as trainIMM
from . import lwf as trainLWF
from . import mas as trainMAS
from . import packnet_main as trainPacknet
from . import si as trainSI
from . import train_common as tc


class Category(Enum):
    MODEL_BASED = auto()
    DATA_BASED = auto()
    MASK_BASED = auto()
    BASELINE = auto()
    REHEARSAL_BASED = auto()


class Method(ABC):
    @property
    @abstractmethod
    def name(self): pass

    @property
    @abstractmethod
    def eval_name(self): pass

    @property
    @abstractmethod
    def category(self): pass

    @property
    @abstractmethod
    def extra_hyperparams_count(self): pass

    @property
    @abstractmethod
    def hyperparams(self): pass

    @abstractmethod
    def get_output(self, images, args): pass

    @staticmethod
    @abstractmethod
    def inference_eval(args, manager): pass


def get_output_def(model, heads, images, current_head_idx, final_layer_idx):
    """method.py:230-235: swap in the task head, eval-mode forward (HIP kernels via model.forward)."""
    head = heads[current_head_idx]
    model.classifier._modules[final_layer_idx] = head
    model.eval()
    with torch.no_grad():
        return model(images)


def set_hyperparams(method, hyperparams, static_params=False):
    """'a,b;c,d' grammar of method.py:238-274.  (The reference crashes on a single value such as
    '400' — SURVEY §8 gotcha 10; here a single value is accepted as that value.)"""
    assert isinstance(hyperparams, str)

    def leave_default(x):
        return x == "def" or x == ""
    vals = []
    split_lists = [x.strip() for x in hyperparams.split(";") if len(x) > 0]
    for split_list in split_lists:
        sp = [float(x) for x in split_list.split(",") if not leave_default(x)]
        sp = sp[0] if len(sp) == 1 else sp
        if len(split_lists) == 1:
            vals = sp if isinstance(sp, list) else [sp]
        else:
            vals.append(sp)
    target = getattr(method, "static_hyperparams", None) if static_params else method.hyperparams
    if target is None:
        return
    for idx, (key, _) in enumerate(list(target.items())):
        if idx < len(vals) and not leave_default(vals[idx]):
            target[key] = vals[idx]
    method.init_hyperparams = copy.deepcopy(target)


class ConcatTasks(TensorTaskDataset):
    """ConcatDatasetDynamicLabels (data/imgfolder.py): labels of task j are offset by the class
    counts of tasks < j."""

    def __init__(self, dsets, classes_len):
        xs, ys, off = [], [], 0
        for d, n in zip(dsets, classes_len):
            xs.append(d.x)
            ys.append(d.y + off)
            off += n
        super().__init__(torch.cat(xs), torch.cat(ys), sum((list(d.classes) for d in dsets), []))


class Finetune(Method):
    name = "finetuning"
    eval_name = name
    category = Category.BASELINE
    extra_hyperparams_count = 0
    hyperparams = {}
    grid_chkpt = True
    start_scratch = True
    no_framework = True     # intended path (SURVEY §8 gotcha 12)

    def get_output(self, images, args):
        return get_output_def(args.model, args.heads, images, args.current_head_idx, args.final_layer_idx)

    @staticmethod
    def grid_train(args, manager, lr):
        dataset_path = manager.current_task_dataset_path
        if not isinstance(dataset_path, list):
            dataset_path = [dataset_path]
        dset_dataloader, cumsum_dset_sizes, dset_classes = Finetune.compose_dataset(dataset_path, args.batch_size,
                                                                                     getattr(args, "device", "cuda"))
        return trainFT.fine_tune_SGD(dset_dataloader, cumsum_dset_sizes, dset_classes,
                                     model_path=manager.previous_task_model_path, exp_dir=manager.gridsearch_exp_dir,
                                     num_epochs=args.num_epochs, lr=lr, weight_decay=args.weight_decay,
                                     enable_resume=True, save_models_mode=True, replace_last_classifier_layer=True,
                                     freq=args.saving_freq, device=getattr(args, "device", "cuda"),
                                     batch_size=args.batch_size)

    @staticmethod
    def grid_poststep(args, manager):
        manager.previous_task_model_path = os.path.join(manager.best_exp_grid_node_dirname, "best_model.pth.tar")
        exp_dir = os.path.join(manager.parent_exp_dir, "task_" + str(args.task_counter), "TASK_TRAINING")
        if os.path.islink(exp_dir) or os.path.exists(exp_dir):
            os.unlink(exp_dir)
        os.symlink(os.path.relpath(manager.best_exp_grid_node_dirname, os.path.dirname(exp_dir)), exp_dir)

    @staticmethod
    def compose_dataset(dataset_path, batch_size, device="cuda"):
        """method.py:1057-1079 with DeviceLoader in place of DataLoader(num_workers=4)."""
        imgf = {x: [] for x in ["train", "val"]}
        classes = {x: [] for x in ["train", "val"]}
        sizes = {x: [] for x in ["train", "val"]}
        for p in dataset_path:
            w = load_task_datasets(p, device)
            for mode in ["train", "val"]:
                imgf[mode].append(w[mode])
                classes[mode].append(w[mode].classes)
                sizes[mode].append(len(w[mode]))
        cumsum = {m: sum(sizes[m]) for m in sizes}
        clen = {m: [len(c) for c in classes[m]] for m in classes}
        loaders = {x: DeviceLoader(ConcatTasks(imgf[x], clen[x]) if len(imgf[x]) > 1 else imgf[x][0],
                                   batch_size, True, device) for x in ["train", "val"]}
        return loaders, cumsum, classes

    @staticmethod
    def inference_eval(args, manager):
        """method.py:1081-1103."""
        from ..framework import inference as test_network
        model = tc.load_model(args.eval_model_path)
        if isinstance(model, dict):
            model = model["model"]
        head_layer_idx = str(len(model.classifier._modules) - 1)
        assert isinstance(model.classifier._modules[head_layer_idx], torch.nn.Linear), "NO VALID HEAD IDX"
        target_heads = test_network.get_prev_heads(args.head_paths, head_layer_idx, getattr(args, "device", "cuda"))
        assert len(target_heads) == 1
        return test_network.test_model(manager.method, model, args.dset_path, 0, subset=args.test_set,
                                       target_head=target_heads, batch_size=args.batch_size,
                                       task_idx=args.eval_dset_idx, device=getattr(args, "device", "cuda"))


class _Regularised(Method):
    category = Category.MODEL_BASED
    extra_hyperparams_count = 1

    @staticmethod
    def grid_train(args, manager, lr):
        return Finetune.grid_train(args, manager, lr)

    def get_output(self, images, args):
        return get_output_def(args.model, args.heads, images, args.current_head_idx, args.final_layer_idx)

    @staticmethod
    def inference_eval(args, manager):
        return Finetune.inference_eval(args, manager)


class EWC(_Regularised):
    name = "EWC"
    eval_name = name
    hyperparams = OrderedDict({"lambda": 400})

    def train(self, args, manager, hyperparams):
        return trainEWC.fine_tune_EWC_acuumelation(
            dataset_path=manager.current_task_dataset_path, previous_task_model_path=manager.previous_task_model_path,
            exp_dir=manager.heuristic_exp_dir, data_dir=args.data_dir, reg_sets=manager.reg_sets,
            reg_lambda=hyperparams["lambda"], batch_size=args.batch_size, num_epochs=args.num_epochs, lr=args.lr,
            weight_decay=args.weight_decay, saving_freq=args.saving_freq, device=getattr(args, "device", "cuda"))


class MAS(_Regularised):
    name = "MAS"
    eval_name = name
    hyperparams = OrderedDict({"lambda": 3})

    def train(self, args, manager, hyperparams):
        return trainMAS.fine_tune_objective_based_acuumelation(
            dataset_path=manager.current_task_dataset_path, previous_task_model_path=manager.previous_task_model_path,
            init_model_path=args.init_model_path, exp_dir=manager.heuristic_exp_dir, data_dir=args.data_dir,
            reg_sets=manager.reg_sets, reg_lambda=hyperparams["lambda"], batch_size=args.batch_size,
            weight_decay=args.weight_decay, num_epochs=args.num_epochs, lr=args.lr, norm="L2", b1=False,
            saving_freq=args.saving_freq, device=getattr(args, "device", "cuda"))


class SI(_Regularised):
    name = "SI"
    eval_name = name
    hyperparams = OrderedDict({"lambda": 400})

    def train(self, args, manager, hyperparams):
        return trainSI.fine_tune_elastic(
            dataset_path=manager.current_task_dataset_path, num_epochs=args.num_epochs,
            exp_dir=manager.heuristic_exp_dir, model_path=manager.previous_task_model_path,
            reg_lambda=hyperparams["lambda"], batch_size=args.batch_size, lr=args.lr, init_freeze=0,
            weight_decay=args.weight_decay, saving_freq=args.saving_freq, device=getattr(args, "device", "cuda"))



class PackNet(Method):
    """method.py:415-556 (MASK_BASED): phase 1 = finetune on the free weights per LR, phase 2 = prune +
    post-prune finetune; one wrapped model with a head per task and uint8 ownership masks."""
    name = "packnet"
    eval_name = name
    category = Category.MASK_BASED
    extra_hyperparams_count = 1
    hyperparams = OrderedDict({"prune_perc_per_layer": 0.9})
    grid_chkpt = True
    start_scratch = True

    def __init__(self):
        self.pruned_savename = None
        self.grid_batch_size = 200          # method.py:524 hardcodes 200 for phase 1 (tests shrink it)

    @staticmethod
    def get_dataset_name(task_name):
        return "survey_TASK_" + task_name

    def train_init(self, args, manager):
        self.pruned_savename = os.path.join(manager.heuristic_exp_dir, "best_model_PRUNED")

    def train(self, args, manager, hyperparams):
        prune_lr = args.lr * 0.1            # method.py:437
        manager.overwrite_args = {
            "weight_decay": args.weight_decay, "train_path": manager.current_task_dataset_path,
            "test_path": manager.current_task_dataset_path, "mode": "prune",
            "dataset": self.get_dataset_name(args.task_name), "loadname": manager.best_finetuned_model_path,
            "post_prune_epochs": 10, "prune_perc_per_layer": hyperparams["prune_perc_per_layer"], "lr": prune_lr,
            "finetune_epochs": args.num_epochs, "cuda": True, "save_prefix": self.pruned_savename,
            "train_bn": args.train_bn, "saving_freq": args.saving_freq, "current_dataset_idx": args.task_counter,
            "batch_size": args.batch_size,
        }
        task_lr_acc = trainPacknet.main(manager.overwrite_args, device=getattr(args, "device", "cuda"))
        return None, task_lr_acc

    def get_output(self, images, args):
        return get_output_def(args.model, args.heads, images, args.current_head_idx, args.final_layer_idx)

    def init_next_task(self, manager):
        assert self.pruned_savename is not None
        if os.path.exists(self.pruned_savename + "_final.pth.tar"):
            manager.previous_task_model_path = self.pruned_savename + "_final.pth.tar"
        elif os.path.exists(self.pruned_savename + "_postprune.pth.tar"):
            warnings.warn("Final file not found(no final file saved if finetune gives no improvement)! Using postprune")
            manager.previous_task_model_path = self.pruned_savename + "_postprune.pth.tar"
        else:
            raise Exception("Previous task pruned model final/postprune non-existing: {}".format(self.pruned_savename))

    def grid_prestep(self, args, manager):
        manager.dataset_name = self.get_dataset_name(args.task_name)
        manager.disable_pruning_mask = False
        if args.task_counter == 1:
            init_wrapper_model_name = os.path.join(manager.ft_parent_exp_dir, manager.base_model.name + "_INIT_WRAPPED.pth")
            if not os.path.exists(init_wrapper_model_name):
                arch = "alexnet" if "alexnet" in manager.base_model.name.lower() else "VGGslim_nopretrain"
                trainPacknet.main({
                    "arch": arch, "init_dump": True, "cuda": True, "loadname": manager.previous_task_model_path,
                    "save_prefix": init_wrapper_model_name, "last_layer_idx": manager.base_model.last_layer_idx,
                    "current_dataset_idx": args.task_counter})
            manager.previous_task_model_path = init_wrapper_model_name
            manager.disable_pruning_mask = True        # method.py:505: task 1 trains every weight

    def grid_train(self, args, manager, lr):
        ft_savename = os.path.join(manager.gridsearch_exp_dir, "best_model")
        overwrite_args = {
            "weight_decay": args.weight_decay, "disable_pruning_mask": manager.disable_pruning_mask,
            "train_path": manager.current_task_dataset_path, "test_path": manager.current_task_dataset_path,
            "mode": "finetune", "dataset": manager.dataset_name,
            "num_outputs": len(manager.dataset.classes_per_task[args.task_name]),
            "loadname": manager.previous_task_model_path, "lr": lr, "finetune_epochs": args.num_epochs, "cuda": True,
            "save_prefix": ft_savename, "batch_size": self.grid_batch_size,
            "train_bn": args.train_bn, "saving_freq": args.saving_freq, "current_dataset_idx": args.task_counter,
        }
        acc = trainPacknet.main(overwrite_args, device=getattr(args, "device", "cuda"))
        return None, acc

    def grid_poststep(self, args, manager):
        manager.best_finetuned_model_path = os.path.join(manager.best_exp_grid_node_dirname, "best_model.pth.tar")

    @staticmethod
    def train_args_overwrite(args):
        args.train_bn = "BN" in args.model_name         # ModelRegularization.batchnorm
        print("TRAINING BN PARAMS = ", str(args.train_bn))

    @staticmethod
    def inference_eval(args, manager):
        task_name = manager.dataset.get_taskname(args.eval_dset_idx + 1)
        return trainPacknet.main({
            "train_path": args.dset_path, "test_path": args.dset_path, "mode": "eval",
            "dataset": PackNet.get_dataset_name(task_name), "loadname": args.eval_model_path, "cuda": True,
            "batch_size": args.batch_size, "current_dataset_idx": args.eval_dset_idx + 1},
            device=getattr(args, "device", "cuda"))



def _modular_accespoint(args, manager, parameter, method_arg, save_path=None, prev_model_path=None, finetune=False):
    """method.py:630-660."""
    nc_per_task = [len(v) for v in manager.dataset.classes_per_task.values()]       # dataset_utils.get_nc_per_task
    save_path = manager.heuristic_exp_dir if save_path is None else save_path
    prev_model_path = manager.previous_task_model_path if prev_model_path is None else prev_model_path
    manager.overwrite_args = {
        "weight_decay": args.weight_decay, "task_name": args.task_name, "task_count": args.task_counter,
        "prev_model_path": prev_model_path, "model_name": args.model_name, "output": save_path,
        "nepochs": args.num_epochs, "parameter": parameter, "cuda": True,
        "dataset_path": manager.current_task_dataset_path, "dataset": manager.dataset,
        "n_tasks": manager.dataset.task_count, "batch_size": args.batch_size, "lr": args.lr,
        "is_scratch_model": args.task_counter == 1, "approach": method_arg, "nc_per_task": nc_per_task,
        "finetune_mode": finetune, "save_freq": args.saving_freq,
    }
    return trainHAT.main(manager.overwrite_args, device=getattr(args, "device", "cuda"))


class HAT(Method):
    """method.py:600-627 (MASK_BASED): hard attention to the task; accuracies are fractions in [0, 1]."""
    name = "HAT"
    eval_name = name
    category = Category.MASK_BASED
    extra_hyperparams_count = 2
    hyperparams = OrderedDict({"smax": 800, "c": 2.5})
    start_scratch = True

    def grid_train(self, args, manager, lr):
        args.lr = lr
        return _modular_accespoint(args, manager, list(self.hyperparams.values()), "hat",
                                   save_path=manager.gridsearch_exp_dir, finetune=True)

    def train(self, args, manager, hyperparams):
        return _modular_accespoint(args, manager, list(hyperparams.values()), "hat")

    def get_output(self, images, args):
        from . import hat as H
        head = args.heads[args.current_head_idx]
        eng = getattr(args, "_hat_engine", None)
        if eng is None or eng.net is not args.model or eng.net.classifier[0] is not head:
            args.model.classifier = torch.nn.ModuleList([head])
            eng = args._hat_engine = H.HatEngine(args.model, args.batch_size, tuple(images.shape[1:]), images.device)
        return eng.forward(args.task_idx, images, args.model.smax)

    @staticmethod
    def inference_eval(args, manager):
        return Finetune.inference_eval(args, manager)



def _rehearsal_accespoint(args, manager, memory_strength, mem_per_task, method_arg, save_path=None, prev_model_path=None,
                          finetune=False, postprocess=False):
    """method.py:381-412."""
    nc_per_task = [len(v) for v in manager.dataset.classes_per_task.values()]
    total_outputs = sum(nc_per_task)
    save_path = manager.heuristic_exp_dir if save_path is None else save_path
    prev_model_path = manager.previous_task_model_path if prev_model_path is None else prev_model_path
    manager.overwrite_args = {
        "weight_decay": args.weight_decay, "task_name": args.task_name, "task_count": args.task_counter,
        "prev_model_path": prev_model_path, "save_path": save_path, "n_outputs": total_outputs, "method": method_arg,
        "n_memories": mem_per_task, "n_epochs": args.num_epochs, "memory_strength": memory_strength, "cuda": True,
        "dataset_path": manager.current_task_dataset_path, "n_tasks": manager.dataset.task_count,
        "batch_size": args.batch_size, "lr": args.lr, "finetune": finetune,
        "is_scratch_model": args.task_counter == 1, "postprocess": postprocess,
    }
    return trainRehearsal.main(manager.overwrite_args, nc_per_task, device=getattr(args, "device", "cuda"))


class GEM(Method):
    """method.py:281-327 (REHEARSAL_BASED). Task 1 only wraps the shared SI model with its exemplars."""
    name = "GEM"
    eval_name = name
    category = Category.REHEARSAL_BASED
    extra_hyperparams_count = 1
    hyperparams = OrderedDict({"margin": 1})
    static_hyperparams = OrderedDict({"mem_per_task": 1024})
    wrap_first_task_model = True

    def train(self, args, manager, hyperparams):
        return _rehearsal_accespoint(args, manager, hyperparams["margin"], self.static_hyperparams["mem_per_task"], "gem")

    def get_output(self, images, args):
        offset1, offset2 = args.model.compute_offsets(args.current_head_idx, args.model.cum_nc_per_task)
        return args.model(images, args.current_head_idx)[:, offset1:offset2]

    def poststep(self, args, manager):
        if args.task_counter > 1:
            return
        start_time = time.time()
        save_path = manager.best_model_path
        prev_model_path = manager.previous_task_model_path
        if not os.path.exists(save_path):
            args.lr = getattr(args, "lr", None) or 0.0           # the wrapper's optimizer is rebuilt at the next task
            _rehearsal_accespoint(args, manager, self.hyperparams["margin"], self.static_hyperparams["mem_per_task"],
                                  "gem", save_path, prev_model_path, postprocess=args.task_counter == 1)
        args.postprocess_time = time.time() - start_time
        manager.best_model_path = save_path

    def grid_train(self, args, manager, lr):
        args.lr = lr
        return _rehearsal_accespoint(args, manager, 0, self.static_hyperparams["mem_per_task"], "gem",
                                     save_path=manager.gridsearch_exp_dir, finetune=True)

    @staticmethod
    def inference_eval(args, manager):
        """FinetuneRehearsalFullMem.inference_eval (method.py:1172-1182)."""
        from ..framework import inference as test_network
        model = tc.load_model(args.eval_model_path)
        return test_network.test_model(manager.method, model, args.dset_path, args.eval_dset_idx, subset=args.test_set,
                                       target_head=None, batch_size=args.batch_size, task_idx=args.eval_dset_idx,
                                       device=getattr(args, "device", "cuda"))



class IMM(Method):
    """method.py:760-819 (MODEL_BASED, no_framework): L2-transfer training per task, mean / mode merge before eval."""
    name = "IMM"
    eval_name = name
    modes = ["mean", "mode"]
    category = Category.MODEL_BASED
    extra_hyperparams_count = 1
    hyperparams = OrderedDict({"lambda": 0.01})
    grid_chkpt = True
    no_framework = True

    def __init__(self, mode="mode"):
        if mode not in self.modes:
            raise Exception("NO EXISTING IMM MODE: '{}'".format(mode))
        self.mode = mode
        self.eval_name = self.name + "_" + self.mode

    def set_mode(self, mode):
        if mode not in self.modes:
            raise Exception("TRY TO SET NON EXISTING IMM MODE: ", mode)
        self.mode = mode
        self.eval_name = self.name + "_" + self.mode

    def grid_train(self, args, manager, lr):
        return trainIMM.fine_tune_l2transfer(dataset_path=manager.current_task_dataset_path,
                                             model_path=manager.previous_task_model_path,
                                             exp_dir=manager.gridsearch_exp_dir, reg_lambda=self.hyperparams["lambda"],
                                             batch_size=args.batch_size, num_epochs=args.num_epochs, lr=lr,
                                             weight_decay=args.weight_decay, saving_freq=args.saving_freq,
                                             device=getattr(args, "device", "cuda"))

    def get_output(self, images, args):
        return get_output_def(args.model, args.heads, images, args.current_head_idx, args.final_layer_idx)

    @staticmethod
    def grid_poststep(args, manager):
        Finetune.grid_poststep(args, manager)

    def eval_model_preprocessing(self, args):
        return trainIMM.preprocess_merge_IMM(self, args.models_path, args.datasets_path, args.batch_size, overwrite=True,
                                             device=getattr(args, "device", "cuda"))

    @staticmethod
    def inference_eval(args, manager):
        return Finetune.inference_eval(args, manager)



class LWF(Method):
    """method.py:940-989 (DATA_BASED): new head per task, old heads distilled from the previous model."""
    name = "LWF"
    eval_name = name
    category = Category.DATA_BASED
    extra_hyperparams_count = 1
    hyperparams = OrderedDict({"lambda": 10})

    def __init__(self, warmup_step=False):
        if warmup_step:
            raise NotImplementedError("LwF head warm-up (fine_tune_freeze) is not on the HIP path")
        self.warmup_step = warmup_step

    @staticmethod
    def grid_train(args, manager, lr):
        return Finetune.grid_train(args, manager, lr)

    def train(self, args, manager, hyperparams):
        return trainLWF.fine_tune_SGD_LwF(dataset_path=manager.current_task_dataset_path,
                                          previous_task_model_path=manager.previous_task_model_path,
                                          init_model_path=getattr(args, "init_model_path", ""),
                                          exp_dir=manager.heuristic_exp_dir, batch_size=args.batch_size,
                                          num_epochs=args.num_epochs, lr=args.lr, init_freeze=0,
                                          weight_decay=args.weight_decay,
                                          last_layer_name=args.classifier_heads_starting_idx,
                                          saving_freq=args.saving_freq, reg_lambda=hyperparams["lambda"],
                                          device=getattr(args, "device", "cuda"))

    def get_output(self, images, args):
        with torch.no_grad():
            outputs = args.model(images)
        if isinstance(outputs, list):
            outputs = outputs[args.current_head_idx]
        return outputs

    @staticmethod
    def inference_eval(args, manager):
        if args.trained_model_idx > 0:
            return GEM.inference_eval(args, manager)        # FinetuneRehearsalFullMem.inference_eval: model as is, head idx
        return Finetune.inference_eval(args, manager)        # the shared SI first-task model


class EBLL(Method):
    """method.py:822-936 (DATA_BASED): LwF + a code loss through one under-complete autoencoder per finished task."""
    name = "EBLL"
    eval_name = name
    category = Category.DATA_BASED
    extra_hyperparams_count = 2
    hyperparams = OrderedDict({"reg_lambda": 10, "ebll_reg_alpha": 1})
    static_hyperparams = OrderedDict({"autoencoder_lr": [0.01], "autoencoder_epochs": 50,
                                      "encoder_alphas": [1e-1, 1e-2], "encoder_dims": [100, 300]})

    @staticmethod
    def grid_train(args, manager, lr):
        return Finetune.grid_train(args, manager, lr)

    def prestep(self, args, manager):
        print("AUTOENCODER PHASE: for prev task ", args.task_counter - 1)
        manager.autoencoder_model_path = self._autoencoder_grid(args, manager)
        print("AUTOENCODER PHASE DONE")

    def _autoencoder_grid(self, args, manager):
        """method.py:842-908: (dim, alpha, lr) grid of autoencoders on the previous task, best by validation accuracy of
        the previous model's classifier on the reconstructed features; checkpointed per grid node."""
        import itertools
        import shutil
        parent = os.path.join(manager.parent_exp_dir, "task_" + str(args.task_counter - 1), "ENCODER_TRAINING")
        processed = {"header": ("dim", "alpha", "lr")}
        ckpt = os.path.join(parent, "grid_checkpoint.pth")
        if os.path.exists(ckpt):
            processed = torch.load(ckpt, weights_only=False)
        best_path, best_acc = None, 0
        for it in itertools.product(self.static_hyperparams["encoder_dims"], self.static_hyperparams["encoder_alphas"],
                                    self.static_hyperparams["autoencoder_lr"]):
            dim, alpha, lr = it
            exp_dir = os.path.join(parent, "dim={}_alpha={}_lr={}".format(str(dim), str(alpha), lr))
            if it in processed:
                acc = processed[it]
            else:
                os.makedirs(exp_dir, exist_ok=True)
                t0 = time.time()
                _, acc = trainEBLL.fine_tune_Adam_Autoencoder(dataset_path=args.previous_task_dataset_path,
                                                              previous_task_model_path=manager.previous_task_model_path,
                                                              exp_dir=exp_dir, batch_size=args.batch_size,
                                                              num_epochs=self.static_hyperparams["autoencoder_epochs"], lr=lr,
                                                              alpha=alpha, last_layer_name=args.classifier_heads_starting_idx,
                                                              auto_dim=dim, device=getattr(args, "device", "cuda"))
                args.presteps_elapsed_time += time.time() - t0
                processed[it] = acc
                torch.save(processed, ckpt)
            print("autoencoder acc={}".format(str(acc)))
            if acc > best_acc:
                if best_path is not None:
                    shutil.rmtree(best_path, ignore_errors=True)
                best_acc, best_path = acc, exp_dir
            else:
                shutil.rmtree(exp_dir, ignore_errors=True)
        if best_acc < 0.40:
            print("[WARNING] Auto-encoder grid not sufficient: max attainable acc = {}".format(str(best_acc)))
        return os.path.join(best_path, "best_model.pth.tar")

    def train(self, args, manager, hyperparams):
        return trainEBLL.fine_tune_SGD_EBLL(dataset_path=manager.current_task_dataset_path,
                                            previous_task_model_path=manager.previous_task_model_path,
                                            autoencoder_model_path=manager.autoencoder_model_path,
                                            init_model_path=getattr(args, "init_model_path", ""),
                                            exp_dir=manager.heuristic_exp_dir, batch_size=args.batch_size,
                                            num_epochs=args.num_epochs, lr=args.lr, init_freeze=0,
                                            reg_alpha=hyperparams["ebll_reg_alpha"], weight_decay=args.weight_decay,
                                            saving_freq=args.saving_freq, reg_lambda=hyperparams["reg_lambda"],
                                            device=getattr(args, "device", "cuda"))

    def get_output(self, images, args):
        with torch.no_grad():
            outputs = args.model(images)
        if isinstance(outputs, tuple):                    # (head outputs, codes); the SI first-task model returns logits
            outputs = outputs[0]
        if isinstance(outputs, list):
            outputs = outputs[args.current_head_idx]
        return outputs

    @staticmethod
    def inference_eval(args, manager):
        return LWF.inference_eval(args, manager)


_REGISTRY = {c.name: c for c in (EWC, MAS, SI, Finetune, PackNet, HAT, GEM, IMM, LWF, EBLL)}


def parse(method_name):
    """method.py:35-78 for the methods on the hot path (GEM / PackNet / HAT register themselves when
    their modules are imported)."""
    if IMM.name in method_name:                               # method.py:40-42: modeIMM, meanIMM, IMM_mode, IMM_mean
        m = IMM(method_name.replace("_", "").replace(IMM.name, "").strip())
        m.hyperparams = copy.deepcopy(IMM.hyperparams)
        return m
    if method_name in _REGISTRY:
        m = _REGISTRY[method_name]()
        m.hyperparams = copy.deepcopy(type(m).hyperparams)
        return m
    raise NotImplementedError("Method not yet parseable: %r" % method_name)


def register(cls):
    _REGISTRY[cls.name] = cls
    return cls
