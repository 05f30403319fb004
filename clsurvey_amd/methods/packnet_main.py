"""PackNet trainer on the HIP path — mirror of src/methods/packnet/main.py (Manager, init_dump, main)
and packnet/networks.py:ModifiedWrapperModel.

Same entry point as the reference: `main(overwrite_args)` with the keys methods/method.py:435-560
passes (mode finetune | prune | eval, init_dump, loadname, save_prefix, ...), same checkpoint dictionary
('model', 'previous_masks', 'dataset2idx', 'dataset2biases', 'accuracy', 'errors', 'epoch',
'val_beat_counts', 'best_val_acc'), same side-effect files (<prefix>.pth.tar, <prefix>_epoch.pth.tar,
<prefix>_postprune.pth.tar, <prefix>_final.pth.tar, <prefix>.json).

What runs where: one batch = ONE clhip_net_loss_step (fwd + CE + bwd of the wrapped net through the
plan executor) + ONE clhip_packnet_sgd_step over the parameter arena, which fuses the reference's
make_grads_zero -> PacknetSGD.step -> make_pruned_zero (main.py:187-193).  Error meters are device
counters read once per epoch.
"""
import argparse
import json
import os
import warnings

import torch
from ..data import load_task_datasets
import torch.nn as nn

from ..data import DeviceLoader
from ..net import NetEngine
from . import packnet as PK
from .packnet import SparsePruner, PacknetSGD


class View(nn.Module):
    """networks.py:353-358."""

    def __init__(self, *shape):
        super().__init__()
        self.shape = shape

    def forward(self, x):
        return x.view(*self.shape)


class ModifiedWrapperModel(nn.Module):
    """networks.py:10-109: `shared` = features + View + classifier[:-1] (everything prunable),
    `classifiers` = one Linear head per dataset, `classifier` = the active head."""

    def __init__(self, raw_model, classifier_last_layer_idx, input_size):
        super().__init__()
        self.make_model(raw_model, classifier_last_layer_idx)
        self.input_size = input_size

    def make_model(self, raw_model, classifier_last_layer_idx, is_pretrained=False):
        self.datasets, self.classifiers = [], nn.ModuleList()
        self.last_layer_in_feats = None
        classifier_input_size = None
        prunable = []
        start_idx = 0
        found = False
        for idx, module in enumerate(raw_model.classifier.children()):
            if not found:
                if idx == start_idx and isinstance(module, nn.Linear):
                    classifier_input_size = module.in_features
                    found = True
                else:
                    start_idx += 1
            if idx != classifier_last_layer_idx:
                prunable.append(module)
            else:
                if not isinstance(module, nn.Linear):
                    raise Exception("Defined hardcoded last layer idx is not Linear")
                self.last_layer_in_feats = module.in_features
                if is_pretrained:
                    self.datasets.append("task_1")
                    self.classifiers.append(module)
        assert isinstance(classifier_input_size, int)
        assert isinstance(self.last_layer_in_feats, int)
        features = list(raw_model.features.children())
        features.append(View(-1, classifier_input_size))
        features.extend(prunable)
        self.shared = nn.Sequential(*features)
        self.classifier = None

    def add_dataset(self, dataset, num_outputs):
        if dataset not in self.datasets:
            self.datasets.append(dataset)
            self.classifiers.append(nn.Linear(self.last_layer_in_feats, num_outputs))

    def set_dataset(self, dataset):
        assert dataset in self.datasets
        self.classifier = self.classifiers[self.datasets.index(dataset)]

    def forward(self, x):
        raise RuntimeError("ModifiedWrapperModel runs through NetEngine (plan_view) on the HIP path")

    def train_nobn(self, mode=True):
        super().train(mode)


class _PlanView(nn.Module):
    """features / classifier view of a wrapper (same module objects) for net.parse_vgg."""

    def __init__(self, wrapper):
        super().__init__()
        mods = list(wrapper.shared.children())
        cut = next(i for i, m in enumerate(mods) if isinstance(m, View))
        self.features = nn.Sequential(*mods[:cut])
        self.classifier = nn.Sequential(*(mods[cut + 1:] + [wrapper.classifier]))


FLAGS = argparse.ArgumentParser()
for _name, _kw in (
        ("--arch", dict(default=None)), ("--mode", dict(choices=["finetune", "prune", "check", "eval"])),
        ("--finetune_layers", dict(choices=["all", "fc", "classifier"], default="all")),
        ("--num_outputs", dict(type=int, default=-1)), ("--last_layer_idx", dict(type=int, default=4)),
        ("--lr", dict(type=float)), ("--lr_decay_every", dict(type=int)), ("--lr_decay_factor", dict(type=float)),
        ("--finetune_epochs", dict(type=int)), ("--batch_size", dict(type=int, default=200)),
        ("--weight_decay", dict(type=float, default=0.0)), ("--dataset", dict(type=str, default="")),
        ("--current_dataset_idx", dict(type=str, default=None)), ("--train_path", dict(type=str, default="")),
        ("--test_path", dict(type=str, default="")), ("--save_prefix", dict(type=str, default="../checkpoints/")),
        ("--loadname", dict(type=str, default="")), ("--prune_method", dict(type=str, default="sparse")),
        ("--prune_perc_per_layer", dict(type=float, default=0.5)), ("--post_prune_epochs", dict(type=int, default=0)),
        ("--disable_pruning_mask", dict(action="store_true", default=False)),
        ("--train_biases", dict(action="store_true", default=False)),
        ("--train_bn", dict(action="store_true", default=False)), ("--cuda", dict(action="store_true", default=True)),
        ("--init_dump", dict(action="store_true", default=False))):
    FLAGS.add_argument(_name, **_kw)


def set_lr(optimizer, lr, count):
    """packnet/utils.py:25-38."""
    continue_training = True
    if count > 10:
        continue_training = False
        print("training terminated")
    if count == 5:
        lr = lr * 0.1
        print("lr is set to {}".format(lr))
        for param_group in optimizer.param_groups:
            param_group["lr"] = lr
    return optimizer, lr, continue_training


class Manager(object):
    """main.py:83-365."""

    def __init__(self, args, model, previous_masks, dataset2idx, dataset2biases, device="cuda"):
        self.args = args
        self.cuda = args.cuda
        self.model = model
        self.device = torch.device(device)
        self.dataset2idx = dataset2idx
        self.dataset2biases = dataset2biases
        self.engine = None
        if args.mode != "check":
            if "survey" not in args.dataset:
                raise NotImplementedError("only the survey dataset format (main.py:95-108) is on this path")
            dsets = load_task_datasets(args.train_path)
            self.train_data_loader = DeviceLoader(dsets["train"], args.batch_size, True, self.device)
            self.test_data_loader = DeviceLoader(dsets["val" if args.mode != "eval" else "test"], args.batch_size, True,
                                                 self.device)
            self.pruner = SparsePruner(self.model, self.args.prune_perc_per_layer, previous_masks,
                                       self.args.train_biases, self.args.train_bn, self.args.current_dataset_idx)
            in_shape = tuple(self.train_data_loader.x.shape[1:])
            self.model.to(self.device)
            self.engine = NetEngine(_PlanView(self.model), args.batch_size, in_shape, self.device)
            self._stats = torch.zeros(2, dtype=torch.float64, device=self.device)
            self._mask_arena = None

    # ---- mask arena for the fused batch tail
    def _build_mask_arena(self):
        """uint8 image of the parameter arena: weights of `shared` carry their PackNet mask, their
        biases 255 unless train_biases (=> grad zeroed, prune.py:91-93), the head the current index."""
        A = self.engine.arena
        cur = self.pruner.current_dataset_idx
        m = torch.full((A.numel,), cur, dtype=torch.uint8, device=self.device)
        for module_idx, module in self.pruner._layers():
            o, n = A.slot(module.weight)
            m[o:o + n] = self.pruner.current_masks[module_idx].view(-1)
            if module.bias is not None and not self.args.train_biases:
                o, n = A.slot(module.bias)
                m[o:o + n] = 255
        self._mask_arena = m

    def _error(self, loader, backward, tail=None):
        """One pass; returns ClassErrorMeter-style [top-1 error %] (main.py:136-148,176-183)."""
        self._stats.zero_()
        n = 0
        for batch, label in loader:
            self.engine.loss_step(batch, label, "ce_mean", backward=backward, stats=self._stats)
            n += batch.shape[0]
            if tail is not None:
                tail()
        return [100.0 * (1.0 - float(self._stats[1].item()) / max(n, 1))]

    def eval(self, dataset_idx, biases=None):
        """main.py:127-160."""
        if not self.args.disable_pruning_mask:
            self.pruner.apply_mask(dataset_idx)
        else:
            warnings.warn("disable_pruning_mask ")
        if biases is not None:
            self.pruner.restore_biases(biases)
        errors = self._error(self.test_data_loader, backward=False)
        print("Error: @1=%.2f" % errors[0])
        return errors

    def do_epoch(self, epoch_idx, optimizer, set_cuda_hack=False, mem_snapshotted=True):
        """main.py:162-206: per batch zero_grad -> fwd/bwd -> foreign grads 0 -> PacknetSGD -> pruned 0."""
        A = self.engine.arena
        group = optimizer.param_groups[0]
        if ParamCheck.arena_covers(A, optimizer):
            st = optimizer.state.setdefault("__arena__", {})
            if "buf" not in st:
                st["buf"] = torch.zeros_like(A.theta)
                st["first"] = True
            mask = None
            if not self.args.disable_pruning_mask:
                if self._mask_arena is None:
                    self._build_mask_arena()
                mask = self._mask_arena

            def tail():
                first = st["first"] or group["momentum"] == 0
                if mask is None:
                    PK.check(PK._lib.lib().clhip_packnet_sgd_step(
                        A.theta.data_ptr(), A.grad.data_ptr(), st["buf"].data_ptr(), None, A.numel, 0,
                        float(group["lr"]), float(group["momentum"]), float(group["weight_decay"]), int(first),
                        PK._stream()), "clhip_packnet_sgd_step")
                else:
                    PK.fused_batch_tail(A.theta, A.grad, st["buf"], mask, self.pruner.current_dataset_idx,
                                        group["lr"], group["momentum"], group["weight_decay"], first)
                st["first"] = False
        else:
            def tail():                     # finetune_layers != 'all': the reference's three separate steps
                if not self.args.disable_pruning_mask:
                    self.pruner.make_grads_zero(cuda=set_cuda_hack)
                optimizer.step()
                if not self.args.disable_pruning_mask:
                    self.pruner.make_pruned_zero()
        errors = self._error(self.train_data_loader, backward=True, tail=tail)
        print("Training Error: @1=%.2f" % errors[0])
        return errors

    def save_model(self, epoch, best_accuracy, errors, savename):
        """main.py:208-232."""
        self.dataset2idx[self.args.dataset] = self.pruner.current_dataset_idx
        self.dataset2biases[self.args.dataset] = self.pruner.get_biases()
        ckpt = {
            "args": self.args, "epoch": epoch, "accuracy": best_accuracy, "errors": errors,
            "dataset2idx": self.dataset2idx, "previous_masks": self.pruner.current_masks, "model": self.model,
            "val_beat_counts": self.args.val_beat_counts, "best_val_acc": self.args.best_val_acc,
        }
        if self.args.train_biases:
            ckpt["dataset2biases"] = self.dataset2biases
        torch.save(ckpt, savename)
        print("SAVED MODEL TO: ", savename)

    def train(self, epochs, optimizer, save=True, savename="", best_accuracy=0, set_cuda_hack=False, survey_mode=True):
        """main.py:234-306. Returns the best top-1 validation accuracy in percent."""
        error_history = []
        epoch_savename = savename + "_epoch.pth.tar"
        savename = savename + ".pth.tar"
        val_beat_counts = self.args.val_beat_counts
        best_val_acc = self.args.best_val_acc
        for idx in range(self.args.starting_epoch, epochs):
            epoch_idx = idx + 1
            if not survey_mode:
                raise NotImplementedError("step_lr schedule of the PackNet paper runs is out of scope")
            optimizer, lr, continue_training = set_lr(optimizer, self.args.lr, count=val_beat_counts)
            if not continue_training:
                print("EARLY STOPPED, {} times unimproved".format(val_beat_counts))
                return best_accuracy
            self.do_epoch(epoch_idx, optimizer, set_cuda_hack=set_cuda_hack)
            val_errors = self.eval(self.pruner.current_dataset_idx)
            error_history.append(val_errors)
            val_acc = 100 - val_errors[0]
            if val_acc < best_val_acc:
                val_beat_counts += 1
            else:
                val_beat_counts = 0
                best_val_acc = val_acc
            with open(savename.replace(".pth", "").replace(".tar", "") + ".json", "w") as fout:
                json.dump({"error_history": error_history,
                           "args": {k: v for k, v in vars(self.args).items()
                                    if isinstance(v, (int, float, str, bool, type(None)))}}, fout)
            if val_acc > best_accuracy:
                print("Best model so far, Accuracy: %0.2f%% -> %0.2f%%" % (best_accuracy, val_acc))
                best_accuracy = val_acc
                if save:
                    self.save_model(epoch_idx, best_accuracy, val_errors, savename)
            if epoch_idx % self.args.saving_freq == 0:
                self.save_model(epoch_idx, best_accuracy, val_errors, epoch_savename)
        print("Finished finetuning...")
        print("Best error/accuracy: %0.2f%%, %0.2f%%" % (100 - best_accuracy, best_accuracy))
        return best_accuracy

    def prune(self):
        """main.py:308-339."""
        print("Pre-prune eval:")
        self.eval(self.pruner.current_dataset_idx)
        self.pruner.prune()
        self._mask_arena = None
        self.check(True)
        print("\nPost-prune eval:")
        errors = self.eval(self.pruner.current_dataset_idx)
        accuracy = 100 - errors[0]
        self.save_model(-1, accuracy, errors, self.args.save_prefix + "_postprune.pth.tar")
        if self.args.post_prune_epochs:
            print("Doing some extra finetuning...")
            # main.py:326 passes model.parameters(); the heads of other datasets never receive gradients, so
            # the parameters that move are exactly the arena's (shared + active head)
            optimizer = PacknetSGD(self.engine.arena.params, lr=self.args.lr, momentum=0.9,
                                   weight_decay=self.args.weight_decay)
            accuracy = self.train(self.args.post_prune_epochs, optimizer, save=True,
                                  savename=self.args.save_prefix + "_final", best_accuracy=accuracy, set_cuda_hack=True)
        print("Pruning summary:")
        self.check(True)
        return accuracy

    def check(self, verbose=False):
        """main.py:341-351. Returns [(layer_idx, zeros, numel)]."""
        out = []
        for layer_idx, module in enumerate(self.model.shared.modules()):
            if isinstance(module, (nn.Conv2d, nn.Linear)):
                weight = module.weight.data
                num_params = weight.numel()
                num_zero = int(weight.view(-1).eq(0).sum().item())
                out.append((layer_idx, num_zero, num_params))
                if verbose:
                    print("Layer #%d: Pruned %d/%d (%.2f%%)" % (layer_idx, num_zero, num_params,
                                                                100.0 * num_zero / num_params))
        return out


class ParamCheck:
    @staticmethod
    def arena_covers(arena, optimizer):
        """True when the optimizer updates exactly the arena's parameters with one group
        (finetune_layers == 'all'), so the batch tail can be the single fused kernel."""
        if len(optimizer.param_groups) != 1:
            return False
        ps = optimizer.param_groups[0]["params"]
        return len(ps) == len(arena.params) and all(a is b for a, b in zip(ps, arena.params))


def init_dump(args):
    """main.py:354-413 (survey architectures only)."""
    arch = args.arch
    if arch not in ("VGGslim_nopretrain", "VGGslim_trained_first_task", "alexnet"):
        raise ValueError("Architecture type not supported.")
    raw_model = torch.load(args.loadname, weights_only=False)
    input_size = 224 if arch == "alexnet" else 64
    model = ModifiedWrapperModel(raw_model, args.last_layer_idx, input_size)
    dataset2idx = {"nopretrain": 1} if arch == "VGGslim_nopretrain" else {args.dataset: 1}
    previous_masks = {}
    for module_idx, module in enumerate(model.shared.modules()):
        if isinstance(module, (nn.Conv2d, nn.Linear)):
            previous_masks[module_idx] = torch.zeros(module.weight.shape, dtype=torch.uint8,
                                                     device=module.weight.device)
    torch.save({"dataset2idx": dataset2idx, "previous_masks": previous_masks, "model": model}, args.save_prefix)


def main(overwrite_args, device="cuda"):
    """main.py:416-546."""
    args = FLAGS.parse_known_args([])[0]
    for key_arg, val_arg in overwrite_args.items():
        setattr(args, key_arg, val_arg)
    if args.init_dump:
        init_dump(args)
        return
    if args.prune_perc_per_layer <= 0:
        return
    assert args.current_dataset_idx, "Need to explicitly pass the task number"
    if not hasattr(args, "saving_freq"):
        args.saving_freq = 10 ** 9

    ft_epoch_savename = args.save_prefix + "_epoch.pth.tar"
    prune_epoch_savename = args.save_prefix + "_final_epoch.pth.tar"
    resume = ft_epoch_savename if os.path.exists(ft_epoch_savename) else \
        prune_epoch_savename if os.path.exists(prune_epoch_savename) else None
    if resume is not None:
        ckpt = torch.load(resume, weights_only=False)
        args.starting_epoch = ckpt["epoch"] + 1
        args.val_beat_counts = ckpt["val_beat_counts"]
        args.best_val_acc = ckpt["best_val_acc"]
        print("STARTING FROM EPOCH {}, val_beat_count {}, best_val_acc {}".format(
            args.starting_epoch, args.val_beat_counts, args.best_val_acc))
    else:
        ckpt = torch.load(args.loadname, weights_only=False)
        args.starting_epoch = 0
        args.val_beat_counts = 0
        args.best_val_acc = 0
    model = ckpt["model"]
    previous_masks = {k: v.to(device) for k, v in ckpt["previous_masks"].items()}
    dataset2idx = ckpt["dataset2idx"]
    dataset2biases = ckpt.get("dataset2biases", {})

    model.add_dataset(args.dataset, args.num_outputs)
    model.set_dataset(args.dataset)
    model = model.to(device)
    manager = Manager(args, model, previous_masks, dataset2idx, dataset2biases, device)

    if args.mode == "finetune":
        manager.pruner.make_finetuning_mask()
        if args.finetune_layers == "all":
            params_to_optimize = manager.engine.arena.params        # shared + active head (the params with grads)
        elif args.finetune_layers == "classifier":
            for param in model.shared.parameters():
                param.requires_grad = False
            params_to_optimize = model.classifier.parameters()
        else:
            raise NotImplementedError("finetune_layers='fc' is specific to the 4096-wide VGG16 of the PackNet paper")
        optimizer = PacknetSGD(params_to_optimize, lr=args.lr, momentum=0.9, weight_decay=args.weight_decay)
        return manager.train(args.finetune_epochs, optimizer, save=True, savename=args.save_prefix)
    elif args.mode == "prune":
        return manager.prune()
    elif args.mode == "check":
        return manager.check(verbose=True)
    elif args.mode == "eval":
        biases = ckpt["dataset2biases"][args.dataset] if "dataset2biases" in ckpt else None
        print("TESTING ON DATASET IDX {} in {}".format(args.dataset, ckpt["dataset2idx"]))
        idx = ckpt["dataset2idx"][args.dataset]
        eval_errors = manager.eval(int(idx), biases)
        return 100 - eval_errors[0]
