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
import json
import os
import warnings
from types import SimpleNamespace

import torch
import torch.nn as nn

from ..data import DeviceLoader, load_task_datasets
from ..net import NetEngine
from . import packnet as PK
from . import train_common as tc
from .packnet import PacknetSGD, SparsePruner


class View(nn.Module):
    """Reshape as a module: the flatten between `features` and the classifier inside `shared` (networks.py:353-358)."""

    def __init__(self, *shape):
        super().__init__()
        self.shape = shape

    def forward(self, x):
        return x.view(*self.shape)


class ModifiedWrapperModel(nn.Module):
    """The PackNet view of a survey net (networks.py:10-109), same attribute names so checkpoints keep their layout:
      shared       features + View + every classifier module except the head slot   (everything prunable)
      classifiers  one Linear head per dataset (ModuleList, parallel to `datasets`)
      classifier   the active head (set_dataset)"""

    def __init__(self, raw_model, classifier_last_layer_idx, input_size):
        super().__init__()
        self.make_model(raw_model, classifier_last_layer_idx)
        self.input_size = input_size

    def make_model(self, raw_model, classifier_last_layer_idx, is_pretrained=False):
        cls = list(raw_model.classifier.children())
        head = cls[classifier_last_layer_idx]
        if not isinstance(head, nn.Linear):
            raise Exception("Defined hardcoded last layer idx is not Linear")
        first_fc = next((m for m in cls if isinstance(m, nn.Linear)), None)
        assert first_fc is not None, "classifier without a Linear layer"
        self.last_layer_in_feats = head.in_features
        self.datasets, self.classifiers = [], nn.ModuleList()
        if is_pretrained:                       # a model that arrives with a trained first head keeps it as task_1
            self.datasets.append("task_1")
            self.classifiers.append(head)
        body = [m for i, m in enumerate(cls) if i != classifier_last_layer_idx]
        self.shared = nn.Sequential(*list(raw_model.features.children()), View(-1, first_fc.in_features), *body)
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
        """networks.py:86-88: for this wrapper the same as train(); BatchNorm keeps following the batch."""
        super().train(mode)


class _PlanView(nn.Module):
    """features / classifier view of a wrapper (same module objects) for net.parse_vgg."""

    def __init__(self, wrapper):
        super().__init__()
        mods = list(wrapper.shared.children())
        cut = next(i for i, m in enumerate(mods) if isinstance(m, View))
        self.features = nn.Sequential(*mods[:cut])
        self.classifier = nn.Sequential(*(mods[cut + 1:] + [wrapper.classifier]))


# command-line surface of packnet/main.py:29-81; every caller passes a dict that overrides these
DEFAULTS = dict(arch=None, mode=None, finetune_layers="all", num_outputs=-1, last_layer_idx=4, lr=None, lr_decay_every=None,
                lr_decay_factor=None, finetune_epochs=None, batch_size=200, weight_decay=0.0, dataset="",
                current_dataset_idx=None, train_path="", test_path="", save_prefix="../checkpoints/", loadname="",
                prune_method="sparse", prune_perc_per_layer=0.5, post_prune_epochs=0, disable_pruning_mask=False,
                train_biases=False, train_bn=False, cuda=True, init_dump=False)
MODES = ("finetune", "prune", "check", "eval")


class Manager(object):
    """One (task, mode) session over a wrapped net: data on the device, pruner, plan executor (main.py:83-365)."""

    def __init__(self, args, model, previous_masks, dataset2idx, dataset2biases, device="cuda"):
        self.args, self.model = args, model
        self.cuda = args.cuda
        self.device = torch.device(device)
        self.dataset2idx, self.dataset2biases = dataset2idx, dataset2biases
        self.engine = None
        if args.mode == "check":
            return
        if "survey" not in args.dataset:
            raise NotImplementedError("only the survey dataset format (main.py:95-108) is on this path")
        dsets = load_task_datasets(args.train_path)
        held_out = "test" if args.mode == "eval" else "val"
        self.train_data_loader = DeviceLoader(dsets["train"], args.batch_size, True, self.device)
        self.test_data_loader = DeviceLoader(dsets[held_out], args.batch_size, True, self.device)
        self.pruner = SparsePruner(model, args.prune_perc_per_layer, previous_masks, args.train_biases, args.train_bn,
                                   args.current_dataset_idx)
        model.to(self.device)
        self.engine = NetEngine(_PlanView(model), args.batch_size, tuple(self.train_data_loader.x.shape[1:]), self.device)
        self._stats = torch.zeros(2, dtype=torch.float64, device=self.device)
        self._mask_arena = None

    # ------------------------------------------------------------------ fused batch tail
    def _build_mask_arena(self):
        """uint8 image of the parameter arena for clhip_packnet batch tail: a weight of `shared` carries its PackNet
        owner index, everything whose gradient the reference zeroes carries 255 — the shared biases unless train_biases
        (prune.py:91-93) and the BatchNorm scale / shift unless train_bn (prune.py:94-98) — the head the current index."""
        A = self.engine.arena
        m = torch.full((A.numel,), self.pruner.current_dataset_idx, dtype=torch.uint8, device=self.device)

        def fill(p, value):
            o, n = A.slot(p)
            m[o:o + n] = value
        for module_idx, module in self.pruner._layers():
            fill(module.weight, self.pruner.current_masks[module_idx].view(-1))
            if module.bias is not None and not self.args.train_biases:
                fill(module.bias, 255)
        if not self.args.train_bn:
            for module in self.model.shared.modules():
                if isinstance(module, nn.modules.batchnorm._BatchNorm) and module.affine:
                    fill(module.weight, 255)
                    fill(module.bias, 255)
        self._mask_arena = m

    def _mode(self, training):
        """The plan executor reads the mode from the module it was built over — the features / classifier VIEW of the
        wrapper, a module of its own — so both are switched."""
        self.model.train(training)
        self.engine.model.train(training)

    def _sweep(self, loader, backward, after_batch=None):
        """One pass over a loader; top-1 error in percent, ClassErrorMeter style (main.py:136-148, 176-183)."""
        self._stats.zero_()
        seen = 0
        for batch, label in loader:
            self.engine.loss_step(batch, label, "ce_mean", backward=backward, stats=self._stats)
            seen += batch.shape[0]
            if after_batch is not None:
                after_batch()
        return [100.0 * (1.0 - float(self._stats[1].item()) / max(seen, 1))]

    def eval(self, dataset_idx, biases=None):
        """Validation / test error of task `dataset_idx` with only the weights that task may see (main.py:127-162)."""
        if self.args.disable_pruning_mask:
            warnings.warn("disable_pruning_mask ")
        else:
            self.pruner.apply_mask(dataset_idx)
        if biases is not None:
            self.pruner.restore_biases(biases)
        self._mode(False)                       # BatchNorm on its running statistics, Dropout off
        errors = self._sweep(self.test_data_loader, backward=False)
        self._mode(True)                        # main.py:158-161 (train_nobn == train for this wrapper)
        print("Error: @1=%.2f" % errors[0])
        return errors

    def _tail_for(self, optimizer, set_cuda_hack):
        """What follows the backward pass of a training batch: foreign gradients to zero, PacknetSGD, pruned weights to
        zero (main.py:187-193).  One fused kernel when the optimizer steps exactly the arena."""
        A, group = self.engine.arena, optimizer.param_groups[0]
        masked = not self.args.disable_pruning_mask
        ps = group["params"]
        whole_arena = len(optimizer.param_groups) == 1 and len(ps) == len(A.params) and all(a is b for a, b in zip(ps, A.params))
        if not whole_arena:                     # finetune_layers != 'all': the three steps one by one
            def tail():
                if masked:
                    self.pruner.make_grads_zero(cuda=set_cuda_hack)
                optimizer.step()
                if masked:
                    self.pruner.make_pruned_zero()
            return tail
        st = optimizer.state.setdefault("__arena__", {})
        if "buf" not in st:
            st.update(buf=torch.zeros_like(A.theta), first=True)
        if masked and self._mask_arena is None:
            self._build_mask_arena()
        mask = self._mask_arena if masked else None

        def tail():
            first = st["first"] or group["momentum"] == 0
            if mask is None:
                PK.check(PK._lib.lib().clhip_packnet_sgd_step(
                    A.theta.data_ptr(), A.grad.data_ptr(), st["buf"].data_ptr(), None, A.numel, 0, float(group["lr"]),
                    float(group["momentum"]), float(group["weight_decay"]), int(first), PK._stream()), "clhip_packnet_sgd_step")
            else:
                PK.fused_batch_tail(A.theta, A.grad, st["buf"], mask, self.pruner.current_dataset_idx, group["lr"],
                                    group["momentum"], group["weight_decay"], first)
            st["first"] = False
        return tail

    def do_epoch(self, epoch_idx, optimizer, set_cuda_hack=False, mem_snapshotted=True):
        errors = self._sweep(self.train_data_loader, backward=True, after_batch=self._tail_for(optimizer, set_cuda_hack))
        print("Training Error: @1=%.2f" % errors[0])
        return errors

    def save_model(self, epoch, best_accuracy, errors, savename):
        """The checkpoint dictionary of main.py:208-232 (keys are the file format)."""
        args = self.args
        self.dataset2idx[args.dataset] = self.pruner.current_dataset_idx
        self.dataset2biases[args.dataset] = self.pruner.get_biases()
        ckpt = dict(args=args, epoch=epoch, accuracy=best_accuracy, errors=errors, dataset2idx=self.dataset2idx,
                    previous_masks=self.pruner.current_masks, model=self.model, val_beat_counts=args.val_beat_counts,
                    best_val_acc=args.best_val_acc)
        if args.train_biases:
            ckpt["dataset2biases"] = self.dataset2biases
        torch.save(ckpt, savename)
        print("SAVED MODEL TO: ", savename)

    def train(self, epochs, optimizer, save=True, savename="", best_accuracy=0, set_cuda_hack=False, survey_mode=True):
        """Epoch loop of main.py:234-306 on the survey schedule (count-based LR drop / early stop, always restarted from
        args.lr).  Returns the best top-1 validation accuracy in percent."""
        if not survey_mode:
            raise NotImplementedError("step_lr schedule of the PackNet paper runs is out of scope")
        args = self.args
        log_path = savename + ".json"
        history, stale, best_val = [], args.val_beat_counts, args.best_val_acc
        self._mode(True)
        for epoch_idx in range(args.starting_epoch + 1, epochs + 1):
            optimizer, _, go_on = tc.set_lr(optimizer, args.lr, stale)
            if not go_on:
                print("EARLY STOPPED, {} times unimproved".format(stale))
                return best_accuracy
            self.do_epoch(epoch_idx, optimizer, set_cuda_hack=set_cuda_hack)
            val_errors = self.eval(self.pruner.current_dataset_idx)
            history.append(val_errors)
            val_acc = 100 - val_errors[0]
            stale = stale + 1 if val_acc < best_val else 0
            best_val = max(best_val, val_acc)
            with open(log_path, "w") as fout:
                plain = {k: v for k, v in vars(args).items() if isinstance(v, (int, float, str, bool, type(None)))}
                json.dump({"error_history": history, "args": plain}, fout)
            if val_acc > best_accuracy:
                print("Best model so far, Accuracy: %0.2f%% -> %0.2f%%" % (best_accuracy, val_acc))
                best_accuracy = val_acc
                if save:
                    self.save_model(epoch_idx, best_accuracy, val_errors, savename + ".pth.tar")
            if epoch_idx % args.saving_freq == 0:
                self.save_model(epoch_idx, best_accuracy, val_errors, savename + "_epoch.pth.tar")
        print("Finished finetuning...")
        print("Best error/accuracy: %0.2f%%, %0.2f%%" % (100 - best_accuracy, best_accuracy))
        return best_accuracy

    def prune(self):
        """Prune the current task's weights, report, retrain the survivors for post_prune_epochs (main.py:308-339)."""
        args, cur = self.args, self.pruner.current_dataset_idx
        print("Pre-prune eval:")
        self.eval(cur)
        self.pruner.prune()
        self._mask_arena = None                 # ownership changed: rebuild the arena image
        self.check(True)
        print("\nPost-prune eval:")
        errors = self.eval(cur)
        accuracy = 100 - errors[0]
        self.save_model(-1, accuracy, errors, args.save_prefix + "_postprune.pth.tar")
        if args.post_prune_epochs:
            print("Doing some extra finetuning...")
            # the reference hands model.parameters() to the optimizer; heads of other datasets never get gradients, so
            # what moves is exactly the arena (shared + active head)
            optimizer = PacknetSGD(self.engine.arena.params, lr=args.lr, momentum=0.9, weight_decay=args.weight_decay)
            accuracy = self.train(args.post_prune_epochs, optimizer, save=True, savename=args.save_prefix + "_final",
                                  best_accuracy=accuracy, set_cuda_hack=True)
        print("Pruning summary:")
        self.check(True)
        return accuracy

    def check(self, verbose=False):
        """[(layer_idx, zero weights, weights)] per prunable layer (main.py:341-351)."""
        out = []
        for layer_idx, module in enumerate(self.model.shared.modules()):
            if not isinstance(module, (nn.Conv2d, nn.Linear)):
                continue
            total = module.weight.numel()
            zeros = int(module.weight.data.view(-1).eq(0).sum().item())
            out.append((layer_idx, zeros, total))
            if verbose:
                print("Layer #%d: Pruned %d/%d (%.2f%%)" % (layer_idx, zeros, total, 100.0 * zeros / total))
        return out


def init_dump(args):
    """First wrap of a raw net: empty ownership masks, no heads yet (main.py:354-413, survey architectures)."""
    if args.arch not in ("VGGslim_nopretrain", "VGGslim_trained_first_task", "alexnet"):
        raise ValueError("Architecture type not supported.")
    raw_model = torch.load(args.loadname, weights_only=False)
    model = ModifiedWrapperModel(raw_model, args.last_layer_idx, 224 if args.arch == "alexnet" else 64)
    masks = {i: torch.zeros(m.weight.shape, dtype=torch.uint8, device=m.weight.device)
             for i, m in enumerate(model.shared.modules()) if isinstance(m, (nn.Conv2d, nn.Linear))}
    owner = {"nopretrain": 1} if args.arch == "VGGslim_nopretrain" else {args.dataset: 1}
    torch.save({"dataset2idx": owner, "previous_masks": masks, "model": model}, args.save_prefix)


def _starting_point(args):
    """Checkpoint to continue from: an epoch file of an interrupted finetune / post-prune run, else `loadname`."""
    for path in (args.save_prefix + "_epoch.pth.tar", args.save_prefix + "_final_epoch.pth.tar"):
        if os.path.exists(path):
            ckpt = torch.load(path, weights_only=False)
            args.starting_epoch = ckpt["epoch"] + 1
            args.val_beat_counts, args.best_val_acc = ckpt["val_beat_counts"], ckpt["best_val_acc"]
            print("STARTING FROM EPOCH {}, val_beat_count {}, best_val_acc {}".format(
                args.starting_epoch, args.val_beat_counts, args.best_val_acc))
            return ckpt
    args.starting_epoch, args.val_beat_counts, args.best_val_acc = 0, 0, 0
    return torch.load(args.loadname, weights_only=False)


def main(overwrite_args, device="cuda"):
    """Entry point with the reference's contract (main.py:416-546): a dict of overrides in, per mode
    finetune -> best validation accuracy (%), prune -> accuracy after pruning (+ retraining), eval -> test accuracy (%),
    check -> sparsity table; init_dump writes the first wrapper and returns None."""
    args = SimpleNamespace(**{**DEFAULTS, **overwrite_args})
    if args.init_dump:
        return init_dump(args)
    if args.prune_perc_per_layer <= 0:
        return None
    assert args.mode in MODES, args.mode
    assert args.current_dataset_idx, "Need to explicitly pass the task number"
    if not hasattr(args, "saving_freq"):
        args.saving_freq = 10 ** 9

    ckpt = _starting_point(args)
    model = ckpt["model"]
    model.add_dataset(args.dataset, args.num_outputs)
    model.set_dataset(args.dataset)
    manager = Manager(args, model.to(device), {k: v.to(device) for k, v in ckpt["previous_masks"].items()},
                      ckpt["dataset2idx"], ckpt.get("dataset2biases", {}), device)

    if args.mode == "finetune":
        manager.pruner.make_finetuning_mask()
        if args.finetune_layers == "all":
            params = manager.engine.arena.params                    # shared + active head: the tensors with gradients
        elif args.finetune_layers == "classifier":
            for p in model.shared.parameters():
                p.requires_grad = False
            params = model.classifier.parameters()
        else:
            raise NotImplementedError("finetune_layers='fc' is specific to the 4096-wide VGG16 of the PackNet paper")
        optimizer = PacknetSGD(params, lr=args.lr, momentum=0.9, weight_decay=args.weight_decay)
        return manager.train(args.finetune_epochs, optimizer, save=True, savename=args.save_prefix)
    if args.mode == "prune":
        return manager.prune()
    if args.mode == "check":
        return manager.check(verbose=True)
    owner = ckpt["dataset2idx"]
    print("TESTING ON DATASET IDX {} in {}".format(args.dataset, owner))
    biases = ckpt["dataset2biases"][args.dataset] if "dataset2biases" in ckpt else None
    return 100 - manager.eval(int(owner[args.dataset]), biases)[0]
