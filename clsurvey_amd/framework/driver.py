"""Continual Hyperparameter Framework driver — the build's own counterpart of
src/framework/{main,framework_train,lr_grid_train,eval}.py (the reference cannot travel to the GPU
box).  Pure control flow; every tensor op happens behind `manager.method.<hook>`.

Reproduced semantics (file:line of the reference):
  * CLI flags and defaults                                   main.py:17-74
  * task loop, boot LR grid on task 1, Manager fields        main.py:158-220
  * SI first-task model bootstrap path                       main.py:226-241, utils.py:146
  * phase 1: per-LR x finetune_iterations grid, seeding by iteration index, mean-over-iterations
    best-LR rule, StoragePolicy, grid checkpoint             lr_grid_train.py:9-176
  * phase 2: threshold A_ft*(1-p), decay, max attempts, hyperparams.pth.tar + SUCCESS.FLAG
                                                             framework_train.py:76-216
  * eval: seq_res / seq_forgetting dict layout and file name  eval.py:146-247, utils.py:200-230
`--shard` under torch.distributed.run spreads the independent trainings over one process per GPU
(clsurvey_amd.framework.shard): phase-1 grid nodes, speculative phase-2 attempts and the evaluation pairs; models and
metrics move by RCCL broadcast / all_gather, every rank keeps its own results tree.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m clsurvey_amd.framework.driver \
        base_VGG9_cl_512_512 --method_name MAS --synthetic 10,20,8000,2000,1000,64 --shard --test
"""
import argparse
import copy
import operator
import os
import random
import shutil
import sys
import time
import traceback

import numpy as np
import torch

from ..methods import method as methods

RUNMODES = ["first_task_basemodel_dump", "timing_mode", "debug"]


def build_parser():
    p = argparse.ArgumentParser(description="Continual Hyperparameter Framework on MI355X")
    p.add_argument("model_name", type=str)
    p.add_argument("--method_name", type=str, default=None)
    p.add_argument("--ds_name", type=str, default=None)
    p.add_argument("--gridsearch_name", type=str, default="demo")
    p.add_argument("--exp_name", type=str, default=None)
    p.add_argument("--starting_task_count", type=int, default=1)
    p.add_argument("--max_task_count", type=int, default=None)
    p.add_argument("--finetune_iterations", type=int, default=1)
    p.add_argument("--saving_freq", type=int, default=20)
    p.add_argument("--save_models_FT_heuristic", action="store_true")
    p.add_argument("--runmode", default=None, choices=RUNMODES)
    p.add_argument("--cleanup_exp", action="store_true")
    p.add_argument("--drop_margin", type=float, default=0.2)
    p.add_argument("--decaying_factor", type=float, default=0.5)
    p.add_argument("--max_attempts_per_task", type=int, default=10)
    p.add_argument("--hyperparams", type=str, default="")
    p.add_argument("--static_hyperparams", type=str, default="")
    p.add_argument("--lr_grid", type=str, default="1e-2,5e-3,1e-3,5e-4,1e-4")
    p.add_argument("--boot_lr_grid", type=str, default=None)
    p.add_argument("--num_epochs", type=int, default=70)
    p.add_argument("--weight_decay", type=float, default=0)
    p.add_argument("--batch_size", type=int, default=200)
    p.add_argument("--test", action="store_true")
    p.add_argument("--test_max_task_count", type=int, default=None)
    p.add_argument("--test_starting_task_count", type=int, default=1)
    p.add_argument("--test_overwrite_mode", action="store_true")
    p.add_argument("--test_set", choices=["test", "val", "train"], type=str, default="test")
    # build-specific
    p.add_argument("--results_root", type=str, default="./exp_results")
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--shard", action="store_true",
                   help="one process per GPU (torch.distributed.run): grid nodes, decay attempts and evaluations are "
                        "spread over the ranks; each rank writes under <results_root>/rank<r>")
    p.add_argument("--methods", type=str, default=None,
                   help="with --shard: comma-separated method names run SIDE BY SIDE on disjoint rank subsets of the node "
                        "(e.g. MAS,SI on 8 GPUs = 4 + 4; each subset shards its method's grid / decay / evaluation as --shard "
                        "does over the world).  With fewer ranks than methods they run one after the other over all ranks")
    p.add_argument("--no_speculation", action="store_true",
                   help="with --shard: phase 2 runs sequentially on rank 0 only; its model and state are broadcast")
    p.add_argument("--synthetic", type=str, default=None,
                   help="command-line runs: tasks,classes,train,val,test,hw[,noise[,kind[,g,amp,noise_lr,q]]] of a synthetic task "
                        "sequence (clsurvey_amd.framework.tasks; kind = protos | blobs, see data.synthetic_task), "
                        "e.g. 10,20,8000,2000,1000,64 = Tiny-ImageNet's shape")
    return p


def set_random(seed=7):
    """utilities/utils.py:52-58."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    random.seed(seed)
    np.random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)


def float_to_scientific_str(v):
    """utilities/utils.py: one decimal, capital E — 'lr=1.0E-02' is the node directory a reference tree holds (fixture G23)."""
    return "{:.1E}".format(v)


class StoragePolicy(object):
    """lr_grid_train.py:162-176."""

    def __init__(self, mode):
        assert mode in ("all", "only_keep_best", "keep_none")
        self.only_keep_best = mode == "only_keep_best"
        self.keep_none = mode == "keep_none"
        self.keep_all = mode == "all"


class Manager(object):
    """main.py:181-220."""
    token_name = "SUCCESS.FLAG"

    def __init__(self, dataset, method, previous_task_model_path, parent_exp_dir, base_model):
        self.dataset = dataset
        self.method = method
        self.previous_task_model_path = previous_task_model_path
        self.parent_exp_dir = parent_exp_dir
        self.base_model = base_model
        self.current_task_dataset_path = None
        self.best_finetuned_model_path = None
        self.autoencoder_model_path = None
        # (reg_sets appears with the second task, framework_train.py:253, as in the reference)

    def set_dataset(self, args, rnd_transform=False):
        if hasattr(self.method, "grid_datafetch"):
            self.current_task_dataset_path = self.method.grid_datafetch(args, self.dataset)
        else:
            self.current_task_dataset_path = self.dataset.get_task_dataset_path(task_name=args.task_name,
                                                                                rnd_transform=rnd_transform)

    def save_hyperparams(self, output_dir, hyperparams):
        os.makedirs(output_dir, exist_ok=True)
        clean = {k: v for k, v in hyperparams.items() if k not in ("args", "manager")}
        torch.save(clean, os.path.join(output_dir, "hyperparams.pth.tar"))

    def get_success_token_path(self, exp_dir):
        return os.path.join(exp_dir, self.token_name)

    def create_success_token(self, exp_dir):
        if not os.path.exists(self.get_success_token_path(exp_dir)):
            torch.save("", self.get_success_token_path(exp_dir))


class BaseModel(object):
    """models/net.py model wrapper: .name, .path (pickled untrained model), .last_layer_idx."""

    def __init__(self, root, name, input_size, num_classes):
        from .. import models
        self.name = name
        self.path = os.path.join(root, name + ".pth.tar")
        if not os.path.exists(self.path):
            os.makedirs(root, exist_ok=True)
            m = models.parse_model_name(name, input_size, num_classes)
            torch.save(m, self.path)
        m = torch.load(self.path, weights_only=False)
        self.last_layer_idx = len(m.classifier._modules) - 1


# ------------------------------------------------------------------ phase 1
def lr_grid_single_task(args, manager, save_models_mode="keep_none", train_node=None):
    """lr_grid_train.py:9-160. `train_node(lr, iteration) -> acc` lets the shard module run grid
    nodes on other ranks; default runs them here."""
    manager.store_policy = StoragePolicy(save_models_mode)
    args.task_name = manager.dataset.get_taskname(args.task_counter)
    manager.ft_parent_exp_dir = os.path.join(manager.parent_exp_dir, "task_" + str(args.task_counter),
                                             "FT_LR_GRIDSEARCH")
    os.makedirs(manager.ft_parent_exp_dir, exist_ok=True)
    processed_lrs = {}
    grid_checkpoint_file = os.path.join(manager.ft_parent_exp_dir, "grid_checkpoint.pth")
    if os.path.exists(grid_checkpoint_file):
        processed_lrs = torch.load(grid_checkpoint_file, weights_only=False)["processed_lrs"]
    if getattr(manager, "sync_processed", None) is not None:       # sharded grid: one table of finished nodes for all ranks
        processed_lrs = manager.sync_processed(processed_lrs)
    args.presteps_elapsed_time = 0
    if hasattr(manager.method, "grid_prestep"):
        manager.method.grid_prestep(args, manager)

    def node_dir(lr, it):
        d = "lr=" + str(float_to_scientific_str(lr))
        if args.finetune_iterations > 1:
            d += "_it" + str(it)
        return os.path.join(manager.ft_parent_exp_dir, d)

    if train_node is None:
        def train_node(lr, it):
            set_random(it)                                       # lr_grid_train.py:73,77
            manager.gridsearch_exp_dir = node_dir(lr, it)
            os.makedirs(manager.gridsearch_exp_dir, exist_ok=True)
            from . import shard
            with shard.busy("grid", device=getattr(args, "device", None)):      # (stage accounting only: shard.STATS)
                _, acc = manager.method.grid_train(args, manager, lr)
            return acc

    best_acc, best_lr = 0, None
    manager.best_exp_grid_node_dirname = None
    best_iteration_batch_dirs = []
    manager.grid_trace = []
    for lr in args.lrs:
        accum_acc, best_iteration_dir, best_iteration_acc = 0, None, 0
        iteration_batch_dirs = []
        if lr not in processed_lrs:
            processed_lrs[lr] = {"acc": []}
        for it in range(args.finetune_iterations):
            manager.gridsearch_exp_dir = node_dir(lr, it)
            iteration_batch_dirs.append(manager.gridsearch_exp_dir)
            if it < len(processed_lrs[lr]["acc"]):
                acc = processed_lrs[lr]["acc"][it]
                set_random(it)
            else:
                acc = train_node(lr, it)
                processed_lrs[lr]["acc"].append(acc)
            manager.grid_trace.append((lr, it, acc))
            if acc > best_iteration_acc:
                best_iteration_acc, best_iteration_dir = acc, node_dir(lr, it)
            accum_acc += acc
            torch.save({"processed_lrs": processed_lrs}, grid_checkpoint_file)
        avg_acc = accum_acc / args.finetune_iterations
        if avg_acc > best_acc:
            best_lr, best_acc = lr, avg_acc
            manager.best_exp_grid_node_dirname = best_iteration_dir
            if manager.store_policy.only_keep_best:
                for d in best_iteration_batch_dirs:
                    shutil.rmtree(d, ignore_errors=True)
            best_iteration_batch_dirs = iteration_batch_dirs
        elif manager.store_policy.only_keep_best:
            for d in iteration_batch_dirs:
                shutil.rmtree(d, ignore_errors=True)
        if manager.store_policy.keep_none:
            for d in iteration_batch_dirs:
                shutil.rmtree(d, ignore_errors=True)
    print("FINETUNE DONE: best_lr={}, best_acc={}".format(best_lr, best_acc))
    if getattr(manager, "after_grid", None) is not None:         # sharded grid: fetch the winner's files from its rank
        manager.after_grid(args, manager, best_lr)
    if hasattr(manager.method, "grid_poststep"):
        manager.method.grid_poststep(args, manager)
    return best_lr, best_acc


# ------------------------------------------------------------------ phase 2
class HyperparameterFramework(object):
    """framework_train.py:14-216."""

    def __init__(self, method):
        self.hyperparams = method.hyperparams
        self.hyperparams_backup = copy.deepcopy(self.hyperparams)
        self.hyperparam_idx = 0
        self.attempts = 0
        self.trace = []      # (hyperparams, acc, threshold) per attempt — parity observable

    def _get_state(self):
        return {"hyperparams": self.hyperparams, "hyperparams_backup": self.hyperparams_backup,
                "hyperparam_idx": self.hyperparam_idx, "attempts": self.attempts}

    def _restore_state(self, state):
        for hkey in self.hyperparams.keys():
            self.hyperparams[hkey] = state["hyperparams"][hkey]
            self.hyperparams_backup[hkey] = state["hyperparams_backup"][hkey]
        self.hyperparam_idx = state["hyperparam_idx"]
        self.attempts = state["attempts"]

    @staticmethod
    def maximalPlasticitySearch(args, manager, train_node=None):
        t0 = time.time()
        lr, acc = lr_grid_single_task(args, manager, save_models_mode=args.save_models_mode, train_node=train_node)
        args.phase1_elapsed_time = time.time() - t0
        return lr, acc

    def load_chkpt(self, manager):
        os.makedirs(manager.heuristic_exp_dir, exist_ok=True)
        path = os.path.join(manager.heuristic_exp_dir, "hyperparams.pth.tar")
        try:
            chkpt = torch.load(path, weights_only=False)
        except Exception:
            return False
        self._restore_state(chkpt["state"])
        return True

    def stabilityDecay(self, args, manager, finetune_lr, finetune_acc):
        args.lr = finetune_lr
        self.phase1 = (finetune_lr, finetune_acc)                 # (observable only: what phase 2 started from)
        manager.heuristic_exp_dir = os.path.join(manager.parent_exp_dir, "task_" + str(args.task_counter),
                                                 "TASK_TRAINING")
        if hasattr(manager.method, "train_init"):
            manager.method.train_init(args, manager)
        sharded = getattr(manager, "speculative", False)
        collective = sharded or getattr(manager, "sequential_on_rank0", False)
        if collective:
            done = self._resume_decision_from_rank0(manager)
        else:
            if not self.load_chkpt(manager):
                self.attempts = 0
                self.hyperparams_backup = copy.deepcopy(self.hyperparams)
            done = os.path.exists(manager.get_success_token_path(manager.heuristic_exp_dir))
        if done:
            manager.best_model_path = os.path.join(manager.heuristic_exp_dir, "best_model.pth.tar")
            return
        args.presteps_elapsed_time = 0
        if getattr(manager, "sequential_on_rank0", False):
            self._sequential_on_rank0(args, manager, finetune_acc)
            return
        if hasattr(manager.method, "prestep"):
            if sharded:
                self._prestep_on_rank0(args, manager)
            else:
                manager.method.prestep(args, manager)
        if sharded:
            self._speculative_decay(args, manager, finetune_acc)
            return
        self._sequential_decay(args, manager, finetune_acc)

    def _resume_decision_from_rank0(self, manager):
        """Sharded runs: every rank keeps its own results tree, and a run killed between rank 0's success token and the
        broadcast of its task directory (or half-way through that broadcast) leaves trees that DISAGREE about whether this
        task is finished — on resume one rank would return early while the others enter the next collective.  So the
        checkpoint that is restored and the 'already done' decision are rank 0's, for every rank; when the task is done
        rank 0 sends its directory again (the other trees may hold a partial copy; tokens travel last, shard.broadcast_files)."""
        from . import shard
        rank, _ = shard.rank_world()
        mine = None
        if rank == 0:
            loaded = self.load_chkpt(manager)
            if not loaded:
                self.attempts = 0
                self.hyperparams_backup = copy.deepcopy(self.hyperparams)
            # rank 0's state travels in BOTH cases: every rank provably starts from identical hyper-parameters and backup
            mine = {"loaded": loaded, "state": copy.deepcopy(self._get_state()),
                    "done": os.path.exists(manager.get_success_token_path(manager.heuristic_exp_dir))}
        dec = shard.broadcast_object(mine, 0)
        if rank != 0:
            self._restore_state(dec["state"])
        if dec["done"]:
            if rank != 0:
                shutil.rmtree(manager.heuristic_exp_dir, ignore_errors=True)
            shard.broadcast_files(manager.heuristic_exp_dir, 0)
            manager.method.hyperparams = self.hyperparams
        elif rank != 0:
            # rank 0 says the task is NOT finished: nothing of this rank's own earlier attempt (stale token, best_model.pth.tar,
            # hyperparams.pth.tar of a run killed half-way) may survive into the retrain
            shutil.rmtree(manager.heuristic_exp_dir, ignore_errors=True)
            os.makedirs(manager.heuristic_exp_dir, exist_ok=True)
        return dec["done"]

    def _sequential_decay(self, args, manager, finetune_acc):
        """framework_train.py:100-136."""
        max_attempts = args.max_attempts_per_task
        converged = False
        while not converged and self.attempts < max_attempts:
            print(" => ATTEMPT {}/{}: Hyperparams {}".format(self.attempts, max_attempts - 1, self.hyperparams))
            t0 = time.time()
            try:
                manager.method.hyperparams = self.hyperparams
                from . import shard
                with shard.busy("decay", device=getattr(args, "device", None)):
                    model, task_lr_acc = manager.method.train(args, manager, self.hyperparams)
            except Exception:
                traceback.print_exc()
                sys.exit(1)
            threshold = finetune_acc * args.inv_drop_margin          # A_ft * (1 - p)
            self.trace.append((copy.deepcopy(dict(self.hyperparams)), task_lr_acc, threshold))
            if task_lr_acc >= threshold:
                converged = True
                args.convergence_iteration_elapsed_time = time.time() - t0
            else:
                self.hyperparamDecay(args, manager)
                self.attempts += 1
                if self.attempts < max_attempts:
                    shutil.rmtree(manager.heuristic_exp_dir, ignore_errors=True)
                else:
                    converged = True
            manager.save_hyperparams(manager.heuristic_exp_dir,
                                     {"acc_threshold": threshold, "val_acc": task_lr_acc, "state": self._get_state()})
        manager.best_model_path = os.path.join(manager.heuristic_exp_dir, "best_model.pth.tar")
        manager.create_success_token(manager.heuristic_exp_dir)

    def _sequential_on_rank0(self, args, manager, finetune_acc):
        """--shard --no_speculation: prestep and the sequential phase 2 run on rank 0 ONLY; its task directory (model,
        hyperparams, success token) and the framework state are then broadcast, so every rank enters the next task's
        sharded grid from the same model and the same hyper-parameters."""
        from . import shard
        rank, _ = shard.rank_world()
        err = None
        if rank == 0:
            try:
                if hasattr(manager.method, "prestep"):
                    manager.method.prestep(args, manager)
                self._sequential_decay(args, manager, finetune_acc)
            except BaseException as e:                   # incl. the SystemExit of a failed training: reported collectively
                traceback.print_exc()
                err = e
        shard.all_ok(err is None, "sequential phase 2 on rank 0")
        if rank != 0:
            shutil.rmtree(manager.heuristic_exp_dir, ignore_errors=True)
        shard.broadcast_files(manager.heuristic_exp_dir, 0)
        state = shard.broadcast_object({"state": self._get_state(), "trace": self.trace,
                                        "autoencoder": (os.path.relpath(manager.autoencoder_model_path, manager.parent_exp_dir)
                                                        if manager.autoencoder_model_path else None)} if rank == 0 else None, 0)
        if rank != 0:
            self._restore_state(state["state"])
            self.trace = state["trace"]
            if state["autoencoder"]:
                manager.autoencoder_model_path = os.path.join(manager.parent_exp_dir, state["autoencoder"])
                shard.broadcast_files(os.path.dirname(manager.autoencoder_model_path), 0)
        elif state["autoencoder"]:
            shard.broadcast_files(os.path.dirname(manager.autoencoder_model_path), 0)
        manager.method.hyperparams = self.hyperparams
        manager.best_model_path = os.path.join(manager.heuristic_exp_dir, "best_model.pth.tar")

    @staticmethod
    def _prestep_on_rank0(args, manager):
        """A prestep trains something of its own (EBLL: the autoencoder grid on the previous task) and publishes it as
        manager.autoencoder_model_path: trained once, on rank 0, then copied into every rank's tree."""
        from . import shard
        rank, _ = shard.rank_world()
        rel = None
        if rank == 0:
            manager.method.prestep(args, manager)
            if manager.autoencoder_model_path:
                rel = os.path.relpath(manager.autoencoder_model_path, manager.parent_exp_dir)
        rel = shard.broadcast_object(rel, src=0)
        if rel is not None:
            manager.autoencoder_model_path = os.path.join(manager.parent_exp_dir, rel)
            shard.broadcast_files(os.path.dirname(manager.autoencoder_model_path), 0)

    def _speculative_decay(self, args, manager, finetune_acc):
        """Phase 2 with one attempt per rank in flight (shard.speculative_round).  Leaves this object, the trace and
        hyperparams.pth.tar in the state the sequential loop reaches when the same attempts succeed / fail."""
        from . import shard
        threshold = finetune_acc * args.inv_drop_margin
        while self.attempts < args.max_attempts_per_task:      # (a checkpoint restored at the attempt limit trains nothing more)
            t0 = time.time()
            accs, accepted = shard.speculative_round(self, args, manager, finetune_acc)
            upto = accepted if accepted is not None else max(accs)
            for k in range(self.attempts, upto + 1):                 # one trace entry per attempt, in attempt order
                twin = shard.decayed_copy(self, args, manager, k - self.attempts)
                self.trace.append((copy.deepcopy(dict(twin.hyperparams)), accs[k], threshold))
            failures = (upto - self.attempts) + (0 if accs[upto] >= threshold else 1)
            for _ in range(failures):
                self.hyperparamDecay(args, manager)
                self.attempts += 1
            manager.method.hyperparams = self.hyperparams
            if accepted is not None:
                args.convergence_iteration_elapsed_time = time.time() - t0
                manager.save_hyperparams(manager.heuristic_exp_dir, {"acc_threshold": threshold, "val_acc": accs[accepted],
                                                                     "state": self._get_state()})
                break
        manager.best_model_path = os.path.join(manager.heuristic_exp_dir, "best_model.pth.tar")
        manager.create_success_token(manager.heuristic_exp_dir)

    def hyperparamDecay(self, args, manager):
        """framework_train.py:168-216 (single hyperparam; round-robin then joint decay for several)."""
        op = manager.method.decay_operator if hasattr(manager.method, "decay_operator") else operator.mul
        if len(self.hyperparams) == 1:
            hkey = list(self.hyperparams.keys())[0]
            self.hyperparams[hkey] = op(self.hyperparams[hkey], args.decaying_factor)
        elif self.hyperparam_idx == len(self.hyperparams):
            self.hyperparam_idx = 0
            for hkey, hval in self.hyperparams_backup.items():
                self.hyperparams[hkey] = op(hval, args.decaying_factor)
            self.hyperparams_backup = copy.deepcopy(self.hyperparams)
        else:
            hlist = list(self.hyperparams.items())
            hkey = hlist[self.hyperparam_idx][0]
            self.hyperparams[hkey] = op(self.hyperparams_backup[hkey], args.decaying_factor)
            for i, (other, _) in enumerate(hlist):
                if i != self.hyperparam_idx:
                    self.hyperparams[other] = self.hyperparams_backup[other]
            self.hyperparam_idx += 1


def framework_single_task(args, manager, train_node=None):
    """framework_train.py:219-292."""
    if args.task_counter == 1 and not args.train_first_task and not args.wrap_first_task_model:
        print("USING SI AS MODEL FOR FIRST TASK: ", manager.previous_task_model_path)
        return None
    skip_to_post = args.wrap_first_task_model and args.task_counter == 1
    hf = HyperparameterFramework(manager.method)
    if args.save_models_FT_heuristic:
        args.save_models_mode = "all"
    elif manager.method.name == "packnet":
        args.save_models_mode = "only_keep_best"
    else:
        args.save_models_mode = "keep_none"
    args.phase1_elapsed_time = args.presteps_elapsed_time = 0
    args.convergence_iteration_elapsed_time = args.postprocess_time = 0
    if args.task_counter > 1:
        prev = manager.dataset.get_taskname(args.task_counter - 1)
        args.previous_task_dataset_path = manager.dataset.get_task_dataset_path(task_name=prev, rnd_transform=False)
        manager.reg_sets = [args.previous_task_dataset_path]
    args.classifier_heads_starting_idx = manager.base_model.last_layer_idx
    if not skip_to_post:
        ft_lr, ft_acc = hf.maximalPlasticitySearch(args, manager, train_node)
        hf.stabilityDecay(args, manager, ft_lr, ft_acc)
    else:
        # The reference reads manager.best_model_path in GEM.poststep (method.py:307) without ever setting it on
        # this path; the evident intent ("save wrapped SI model in first task best_model_path") is task_1's slot.
        manager.heuristic_exp_dir = os.path.join(manager.parent_exp_dir, "task_" + str(args.task_counter),
                                                 "TASK_TRAINING")
        manager.best_model_path = os.path.join(manager.heuristic_exp_dir, "best_model.pth.tar")
    if hasattr(manager.method, "poststep"):
        manager.method.poststep(args, manager)
    if hasattr(manager.method, "init_next_task"):
        manager.method.init_next_task(manager)
    else:
        manager.previous_task_model_path = manager.best_model_path
    return hf


# ------------------------------------------------------------------ eval
def get_perf_output_filename(method_name, dataset_index):
    return "test_method_performances" + method_name + str(dataset_index) + ".pth"      # utils.py:225-230


def eval_all_models_all_tasks(args, manager, ds_paths, model_paths):
    """eval.py:146-247: model j >= i evaluated on task i with task i's head; forgetting = acc_i(i) - acc_i(j).
    With --shard the (task, model) pairs are spread over the ranks and the accuracies all-gathered."""
    from . import shard
    rank, world = shard.rank_world() if getattr(manager, "speculative", False) or getattr(manager, "after_grid", None) else (0, 1)
    tasks = [i for i in range(args.test_starting_task_count - 1, args.test_max_task_count) if i < len(ds_paths)]
    pairs = [(i, j) for i in tasks for j in range(i, len(ds_paths))]

    def evaluate(dataset_index, trained_model_idx):
        args.eval_dset_idx = dataset_index
        args.dset_path = ds_paths[dataset_index]
        args.head_paths = model_paths[dataset_index]
        args.trained_model_idx = trained_model_idx
        args.eval_model_path = model_paths[trained_model_idx]
        return manager.method.inference_eval(args, manager)

    debug = bool(getattr(args, "debug", False))
    overwrite = bool(getattr(args, "test_overwrite_mode", False))

    def out_file(dataset_index):
        return os.path.join(args.out_path, get_perf_output_filename(manager.method.eval_name, dataset_index))

    def save(dataset_index, seq_acc, seq_forgetting):
        perf = {manager.method.eval_name: {"seq_res": seq_acc, "seq_forgetting": seq_forgetting, "seq_head_acc": []}}
        if not debug:
            os.makedirs(args.out_path, exist_ok=True)
            torch.save(perf, out_file(dataset_index))
        return perf[manager.method.eval_name]

    out = {}
    if shard._solo(world):
        # eval.py:146-247 as it runs: a finished task is never re-evaluated outside overwrite mode (and ends the run), a
        # model whose evaluation fails ends its task's sequence (what was measured is kept), a task with no result at all
        # ends the run; debug mode writes nothing
        for dataset_index in tasks:
            args.eval_dset_idx = dataset_index
            if not overwrite and not debug and os.path.exists(out_file(dataset_index)):
                print("EVAL already done, can only rerun in overwrite mode")
                break
            seq_acc, seq_forgetting = {dataset_index: []}, {dataset_index: []}
            for trained_model_idx in range(dataset_index, len(ds_paths)):
                try:
                    with shard.busy("eval", device=getattr(args, "device", None)):
                        accuracy = evaluate(dataset_index, trained_model_idx)
                except Exception:
                    print("ERROR in Testing model, trained until TASK ", str(trained_model_idx + 1))
                    print("Aborting testing on further models")
                    traceback.print_exc(5)
                    break
                seq_acc[dataset_index].append(accuracy)
                if trained_model_idx > dataset_index:
                    seq_forgetting[dataset_index].append(seq_acc[dataset_index][0] - accuracy)
            if not seq_acc[dataset_index]:
                print("TESTING ERROR: no accuracy for task", dataset_index + 1, "- no results saved")
                break
            out[dataset_index] = save(dataset_index, seq_acc, seq_forgetting)
        return out
    # sharded: every (task, model) pair is independent work; a failing evaluation stops the run on EVERY rank
    if not overwrite and not debug:                                # eval.py:158-162, decided on rank 0's tree for all ranks
        done = shard.broadcast_object([i for i in tasks if os.path.exists(out_file(i))] if rank == 0 else None, 0)
        if done:
            print("EVAL already done, can only rerun in overwrite mode")
            tasks = [i for i in tasks if i < min(done)]
            pairs = [(i, j) for i in tasks for j in range(i, len(ds_paths))]
    table, err = {}, None
    try:
        mine = [(n, pq) for n, pq in enumerate(pairs) if n % world == rank]
        with shard.busy("eval", len(mine), device=getattr(args, "device", None)):
            table = {n: evaluate(*pq) for n, pq in mine}
    except Exception as e:
        traceback.print_exc()
        err = e
    shard.all_ok(err is None, "evaluation")
    table = shard.gather_scalars(table)
    acc_of = {pq: table[n] for n, pq in enumerate(pairs)}
    for dataset_index in tasks:
        seq_acc, seq_forgetting = {dataset_index: []}, {dataset_index: []}
        for trained_model_idx in range(dataset_index, len(ds_paths)):
            accuracy = acc_of[(dataset_index, trained_model_idx)]
            seq_acc[dataset_index].append(accuracy)
            if trained_model_idx > dataset_index:
                seq_forgetting[dataset_index].append(seq_acc[dataset_index][0] - accuracy)
        out[dataset_index] = save(dataset_index, seq_acc, seq_forgetting)
    return out


# ------------------------------------------------------------------ main
def get_exp_name(args, method):
    """utils.py:130-143."""
    parts = ["dm={}".format(args.drop_margin), "df={}".format(args.decaying_factor), "e={}".format(args.num_epochs),
             "bs={}".format(args.batch_size)]
    if args.weight_decay != 0:
        parts.append("L2={}".format(args.weight_decay))
    for k, v in method.hyperparams.items():
        parts.append("{}={}".format(k, v))
    for k, v in getattr(method, "static_hyperparams", {}).items():
        parts.append("{}={}".format(k, v))
    return "_".join(parts)


def first_task_modelname(args):
    """models/net.py:39-53 (get_init_modelname)."""
    name = ["e={}".format(args.num_epochs), "bs={}".format(args.batch_size), "lr={}".format(sorted(args.lr_grid))]
    if args.weight_decay != 0:
        name.append("L2={}".format(args.weight_decay))
    for tag in ("BN", "DROP"):          # a regularised architecture gets its own first-task model (net.py:47-51, this order)
        if tag in args.model_name:
            name.append(tag)
    return "_".join(name)


def main_methods(argv, names, dataset=None):
    """`--shard --methods A,B[,...]` (BASELINE configs 3 / 5 are "MAS + SI" / "PackNet + HAT" on one 8-GPU node):
    the world's ranks are split into one contiguous block per method (shard.split_ranks); each block runs main() for ITS method
    with the block as its sharding world — own communicator, group-relative ranks, nothing exchanged between blocks.  Every
    method's decisions equal those of a single-method run on a world of the block's size (tests/test_shard_gloo.py).  With
    fewer ranks than methods the methods run one after the other, each over all ranks.  Returns {"method": name(s) this rank
    ran, "blocks": {name: ranks}, "out": main()'s dict of this rank's (last) method, "outs": {name: dict}}."""
    from . import shard
    import sys as _sys
    argv = list(_sys.argv[1:] if argv is None else argv)
    clean, skip = [], False
    for a in argv:                                  # drop --methods X / --method_name X: each inner run gets its own
        if skip:
            skip = False
        elif a in ("--methods", "--method_name"):
            skip = True
        elif not (a.startswith("--methods=") or a.startswith("--method_name=")):
            clean.append(a)
    shard.init_from_env()
    grank, gworld = shard.global_rank_world()
    outs = {}
    if gworld >= len(names) and gworld > 1:
        gi, blocks = shard.enter_method_groups(len(names))
        try:
            outs[names[gi]] = main(clean + ["--method_name", names[gi]], dataset=dataset)
        finally:
            shard.leave_method_groups()
        shard.world_barrier()
        blocks = {n: b for n, b in zip(names, blocks)}
        mine = names[gi]
    else:
        for n in names:
            outs[n] = main(clean + ["--method_name", n], dataset=dataset)
        blocks = {n: list(range(gworld)) for n in names}
        mine = list(names)
    return {"method": mine, "blocks": blocks, "out": outs[mine if isinstance(mine, str) else names[-1]], "outs": outs}


def main(argv=None, method=None, dataset=None, train_node_factory=None):
    args = build_parser().parse_args(argv)
    if args.methods:
        names = [n for n in args.methods.split(",") if n]
        if not args.shard:
            raise SystemExit("--methods needs --shard")
        if method is not None or train_node_factory is not None:
            raise ValueError("--methods creates its own method objects")
        return main_methods(argv, names, dataset=dataset)
    speculative = sequential_on_rank0 = False
    if args.shard:
        from . import shard
        rank, world = shard.init_from_env()
        if not shard._solo(world):
            # (the tree is named after the GLOBAL rank: under --methods two blocks both have a block rank 0)
            args.results_root = os.path.join(args.results_root, "rank%d" % shard.global_rank_world()[0])
            train_node_factory = train_node_factory or shard.sharded_grid_factory()
            speculative = not args.no_speculation
            sequential_on_rank0 = args.no_speculation
    if dataset is None and args.synthetic:
        from .tasks import SyntheticTaskSequence
        fields = args.synthetic.split(",")
        n_tasks, n_cls, n_tr, n_va, n_te, hw = [int(v) for v in fields[:6]]
        kind = fields[7] if len(fields) > 7 else "protos"
        blobs = dict(zip(("g", "amp", "noise_lr", "q"), (float(v) for v in fields[8:12]))) or None
        dataset = SyntheticTaskSequence(os.path.join(args.results_root, "data"), task_count=n_tasks, classes_per_task=n_cls,
                                        sizes=(n_tr, n_va, n_te), hw=hw, noise=float(fields[6]) if len(fields) > 6 else 1.0,
                                        kind=kind, blobs=blobs, seed=int(fields[12]) if len(fields) > 12 else 7)
    set_random(7)                                                 # utils.init -> set_random()
    if method is None:
        method = methods.parse(args.method_name)
    assert dataset is not None, "pass a dataset object (clsurvey_amd.framework.tasks)"
    base_model = BaseModel(os.path.join(args.results_root, "models"), args.model_name, dataset.input_size,
                           len(next(iter(dataset.classes_per_task.values()))))
    # main.py:247-252 (init_checks)
    if args.starting_task_count < 1 or args.starting_task_count > dataset.task_count:
        raise ValueError("ERROR: Starting task count should be in appropriate range for dataset! Value = ", args.starting_task_count)
    assert 0 <= args.drop_margin <= 1
    assert 0 <= args.decaying_factor <= 1
    parse_floats = lambda s: [float(x) for x in s.split(",") if x]   # noqa: E731
    args.lr_grid = parse_floats(args.lr_grid)
    args.boot_lr_grid = parse_floats(args.boot_lr_grid) if args.boot_lr_grid else args.lr_grid
    args.data_dir = None
    args.init_model_path = None
    args.max_task_count = dataset.task_count if args.max_task_count is None else args.max_task_count
    args.inv_drop_margin = 1 - args.drop_margin
    args.first_task_modelname = first_task_modelname(args)
    args.train_first_task = bool(getattr(method, "start_scratch", False))
    args.wrap_first_task_model = bool(getattr(method, "wrap_first_task_model", False))
    args.no_framework = bool(getattr(method, "no_framework", False))
    for option in RUNMODES:
        setattr(args, option, args.runmode == option)
    if args.first_task_basemodel_dump:
        assert method.name == "SI", "Define SI method to train first task common model."
        args.train_first_task = True
        args.starting_task_count = args.max_task_count = 1
        args.gridsearch_name = "first_task_basemodel"
        args.exp_name = args.first_task_modelname
    elif args.timing_mode:                                      # main.py:289-300
        args.max_task_count = 4
        args.batch_size = 200
        args.save_models_FT_heuristic = False
        args.finetune_iterations = 1
        args.num_epochs = 10
        args.encoder_dims, args.encoder_alphas, args.autoencoder_epochs = [100], [1e-2], 10
    elif args.debug:                                            # main.py:269-277
        args.finetune_iterations, args.num_epochs, args.saving_freq = 1, 1, 200
        args.batch_size, args.mem_per_task = 200, 20
        # (the reference also writes args.lrs = [0.01] here, main.py:276, which its task loop overwrites with the full
        # grid before anything reads it: a reference debug run trains every LR of the grid, and so does this one)
    if hasattr(method, "train_args_overwrite"):
        method.train_args_overwrite(args)
    methods.set_hyperparams(method, args.hyperparams)
    methods.set_hyperparams(method, args.static_hyperparams, static_params=True)
    if args.exp_name is None:
        args.exp_name = get_exp_name(args, method)
    tr_root = os.path.join(args.results_root, "train")
    parent_exp_dir = os.path.join(tr_root, dataset.train_exp_results_dir, method.name, base_model.name, "gridsearch",
                                  args.gridsearch_name, args.exp_name)
    if args.cleanup_exp and os.path.isdir(parent_exp_dir):
        shutil.rmtree(parent_exp_dir)
    # main.py:226-241
    if args.starting_task_count == 1:
        si_path = os.path.join(tr_root, dataset.train_exp_results_dir, "SI", base_model.name, "gridsearch",
                               "first_task_basemodel", args.first_task_modelname, "task_1", "TASK_TRAINING",
                               "best_model.pth.tar")
        prev = base_model.path if (args.train_first_task or args.first_task_basemodel_dump) else si_path
    else:
        prev = os.path.join(parent_exp_dir, "task_{}".format(args.starting_task_count - 1), "TASK_TRAINING",
                            "best_model.pth.tar")
    if not os.path.exists(prev) and not args.first_task_basemodel_dump:
        raise Exception("NOT EXISTING previous_task_model_path = " + prev)
    if args.first_task_basemodel_dump:                          # main.py:255-263 (check_dump)
        dumped = os.path.join(parent_exp_dir, "task_1", "TASK_TRAINING", "best_model.pth.tar")
        if os.path.exists(dumped):
            raise Exception("Basemodel/link for SI first task already exists!\nNot overwriting, because reference to all "
                            "other methods.\nManually remove model for a new dump:{}".format(dumped))
    manager = Manager(dataset, method, prev, parent_exp_dir, base_model)
    manager.speculative = speculative
    manager.sequential_on_rank0 = sequential_on_rank0
    ds_paths, model_paths, frameworks = [], [], []
    for task_counter in range(args.starting_task_count, args.max_task_count + 1):
        args.task_counter = task_counter
        args.task_name = dataset.get_taskname(task_counter)
        args.lrs = args.boot_lr_grid if task_counter == 1 else args.lr_grid
        manager.set_dataset(args)
        train_node = train_node_factory(args, manager) if train_node_factory else None
        try:
            if args.no_framework:
                lr_grid_single_task(args, manager, save_models_mode="all", train_node=train_node)
            else:
                frameworks.append(framework_single_task(args, manager, train_node))
            ds_paths.append(manager.current_task_dataset_path)
            model_paths.append(manager.previous_task_model_path)
        except RuntimeError as e:
            print("ERROR:", e)
            traceback.print_exc()
            break
    results = None
    if args.test:
        args.test_max_task_count = dataset.task_count if args.test_max_task_count is None else args.test_max_task_count
        exp = args.exp_name if args.test_set == "test" else "{}_{}".format(args.exp_name, args.test_set)
        args.out_path = os.path.join(args.results_root, "test", "results", dataset.test_results_dir, method.eval_name,
                                     base_model.name, args.gridsearch_name, exp)
        if hasattr(method, "eval_model_preprocessing"):          # eval.py:45-46 (IMM: merged models)
            args.models_path, args.datasets_path = model_paths, ds_paths
            model_paths = method.eval_model_preprocessing(args)
        results = eval_all_models_all_tasks(args, manager, ds_paths, model_paths)
    return {"manager": manager, "frameworks": frameworks, "ds_paths": ds_paths, "model_paths": model_paths,
            "results": results, "args": args}


if __name__ == "__main__":
    out = main()
    from . import shard as _shard
    if _shard.rank_world()[0] == 0 and out["results"] is not None:
        for i, r in sorted(out["results"].items()):
            print("task %d: acc %s  forgetting %s" % (i + 1, ["%.2f" % a for a in r["seq_res"][i]],
                                                      ["%.2f" % f for f in r["seq_forgetting"][i]]))
    _shard.barrier()
