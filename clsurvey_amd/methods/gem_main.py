"""GEM trainer on the HIP path — mirror of src/methods/rehearsal/main_rehearsal.py:main (argument handling,
model load / wrap, postprocess = exemplar collection for the first-task model) and
rehearsal/train_rehearsal.py:train_model (epoch / phase loop, count-based LR decay and early stop).

Per training batch: GemNet.observe (memory passes + current pass through clhip_net_loss_step_slice, one
Gram pass, host QP, projection, SGD) or observe_FT in the phase-1 grid; validation through eval_batch.
Loss / hit counters stay on the device and are read once per phase.
"""
import argparse
import copy
import math
import os
import time

import torch
from ..data import load_task_datasets

from ..data import DeviceLoader
from . import gem as G
from .train_common import set_lr


def termination_protocol(since, best_acc, best_model, exp_dir):
    """train_rehearsal.py:35-50."""
    print("Training complete in {:.0f}s, best val Acc: {:4f}".format(time.time() - since, best_acc))
    torch.save(best_model, os.path.join(exp_dir, "best_model.pth.tar"))


def train_model(model, args, dset_sizes, resume="", save_models_mode=False, saving_freq=10):
    """train_rehearsal.py:57-199. Returns (model, best validation accuracy in [0, 1])."""
    optimizer = model.opt
    exp_dir = args.save_path
    lr = args.lr
    num_epochs = args.n_epochs
    since = time.time()
    val_beat_counts = 0
    best_acc = 0.0
    best_model = None
    start_epoch = 0
    if os.path.isfile(resume):
        checkpoint = torch.load(resume, weights_only=False)
        start_epoch = checkpoint["epoch"]
        model.net.load_state_dict(checkpoint["state_dict"])      # in place, by name: parameters and buffers
        optimizer.load_state_dict(checkpoint["optimizer"])
        best_acc, lr, val_beat_counts = checkpoint["best_acc"], checkpoint["lr"], checkpoint["val_beat_counts"]
    os.makedirs(exp_dir, exist_ok=True)
    val_stats = torch.zeros(2, dtype=torch.float64, device=model.device)
    for epoch in range(start_epoch, num_epochs):
        print("Epoch {}/{}".format(epoch, num_epochs - 1))
        for phase in ["train", "val"]:
            if phase == "train":
                optimizer, lr, continue_training = set_lr(optimizer, lr, count=val_beat_counts)
                if not continue_training:
                    termination_protocol(since, best_acc, best_model, exp_dir)
                    return model, best_acc
            running_loss = torch.zeros((), dtype=torch.float64, device=model.device)
            running_corrects = torch.zeros((), dtype=torch.float64, device=model.device)
            projected = []
            val_stats.zero_()
            for inputs, labels in args.dset_loaders[phase]:
                if phase == "train":
                    if args.finetune:
                        loss, correct = model.observe_FT(inputs, args.task_idx, labels)
                    else:
                        loss, correct, batch_stats = model.observe(inputs, args.task_idx, labels)
                        projected.extend(batch_stats["projected_grads"])
                    running_loss = running_loss + loss.double().sum()
                    running_corrects = running_corrects + correct
                else:
                    loss = model.eval_batch(inputs, labels, args.task_idx, val_stats)
                    running_loss = running_loss + loss.double().sum()
            if phase == "val":
                running_corrects = val_stats[1]
            epoch_loss = float(running_loss.item()) / dset_sizes[phase]      # (mean batch losses) / N, as printed by the reference
            epoch_acc = float(running_corrects.item()) / dset_sizes[phase]
            print("{} Loss: {:.4f} Acc: {:.4f}".format(phase, epoch_loss, epoch_acc))
            if projected:
                # one entry per batch, zeros included, as train_rehearsal.py:153-167 prints it; the device counters are
                # stacked and read ONCE per epoch
                dev_counts = [v for v in projected if torch.is_tensor(v)]
                host = iter(torch.stack([v.reshape(()) for v in dev_counts]).cpu().tolist()) if dev_counts else iter(())
                print("projected_grads = {}".format([int(next(host)) if torch.is_tensor(v) else int(v) for v in projected]))
                if hasattr(model, "check_qp_status"):
                    model.check_qp_status()
            if math.isnan(epoch_loss):
                print("Canceling because Nan LOSS")         # train_rehearsal.py:139-141 (checked per phase here)
                return model, best_acc
            if phase == "val":
                if epoch_acc > best_acc:
                    best_acc = epoch_acc
                    if save_models_mode:
                        torch.save(model, os.path.join(exp_dir, "best_model.pth.tar"))
                    val_beat_counts = 0
                    best_model = copy.deepcopy(model)
                    print("-> New best model")
                else:
                    val_beat_counts += 1
        if save_models_mode and epoch % saving_freq == 0:
            torch.save({"epoch": epoch + 1, "lr": lr, "val_beat_counts": val_beat_counts, "epoch_acc": epoch_acc,
                        "best_acc": best_acc, "arch": "alexnet", "model": model, "state_dict": model.net.state_dict(),
                        "optimizer": optimizer.state_dict()}, os.path.join(exp_dir, "epoch.pth.tar"))
    termination_protocol(since, best_acc, best_model, exp_dir)
    return model, best_acc


def main(overwrite_args, nc_per_task, device="cuda"):
    """main_rehearsal.py:69-255 for method == 'gem'."""
    parser = argparse.ArgumentParser()
    for name, kw in (("--task_name", dict(type=str)), ("--task_count", dict(type=int)),
                     ("--prev_model_path", dict(type=str)), ("--save_path", dict(type=str, default="results/")),
                     ("--n_outputs", dict(type=int, default=200)), ("--method", dict(type=str, default="gem")),
                     ("--postprocess", dict(action="store_true")), ("--weight_decay", dict(type=float, default=0)),
                     ("--is_scratch_model", dict(action="store_true")), ("--n_memories", dict(type=int, default=0)),
                     ("--memory_strength", dict(default=0, type=float)), ("--finetune", dict(action="store_true")),
                     ("--n_epochs", dict(type=int, default=1)), ("--batch_size", dict(type=int, default=70)),
                     ("--lr", dict(type=float, default=1e-3)), ("--n_tasks", dict(type=int, default=10))):
        parser.add_argument(name, **kw)
    args = parser.parse_known_args([])[0]
    args.nc_per_task = nc_per_task
    for key_arg, val_arg in overwrite_args.items():
        setattr(args, key_arg, val_arg)
    args.task_idx = args.task_count - 1
    if args.method != "gem":
        raise NotImplementedError("rehearsal method %r (iCaRL / rehearsal baselines are out of scope)" % args.method)
    assert args.n_outputs == sum(args.nc_per_task)
    assert args.n_tasks == len(nc_per_task)
    if args.task_count == 1:
        assert "SI" in args.prev_model_path, "FIRST TASK NOT STARTING FROM SCRATCH, BUT FROM SI: ONLY STORING WRAPPER " \
                                             "WITH EXEMPLARS, path = {}".format(args.prev_model_path)
        assert args.postprocess, "FIRST TASK WE DO ONLY POSTPROCESSING"
    assert os.path.isfile(args.prev_model_path), "Must specify existing prev_model_path, got: " + args.prev_model_path

    dsets = load_task_datasets(args.dataset_path)
    args.task_imgfolders = dsets
    args.dset_loaders = {x: DeviceLoader(dsets[x], args.batch_size, True, device) for x in ["train", "val"]}
    dset_sizes = {x: len(dsets[x]) for x in ["train", "val"]}
    in_shape = tuple(args.dset_loaders["train"].x.shape[1:])

    if args.is_scratch_model:
        assert args.task_idx == 0
        raw = torch.load(args.prev_model_path, weights_only=False)
        raw = G.extend_head(raw, args.n_outputs)                   # gem.py:96-113
        model = G.GemNet(raw, args.n_outputs, args.n_tasks, args.nc_per_task, args.n_memories, args.lr,
                         args.weight_decay, args.memory_strength, args.batch_size, in_shape, device)
    else:
        model = torch.load(args.prev_model_path, weights_only=False)
        if model.batch_size < args.batch_size:
            model.batch_size = args.batch_size
            model._bind()
    model.init_setup(args)
    assert model.n_tasks == args.n_tasks, "model tasks={}, args tasks={}".format(model.n_tasks, args.n_tasks)
    assert model.n_outputs == args.n_outputs

    if args.postprocess:
        model.manage_memory(args.task_idx, args.dset_loaders["train"])
        os.makedirs(os.path.dirname(args.save_path), exist_ok=True)
        torch.save(model, args.save_path)
        print("SAVED POSTPROCESSED MODEL TO: {}".format(args.save_path))
        return None, None
    resume = os.path.join(args.save_path, "epoch.pth.tar")
    return train_model(model, args, dset_sizes, resume=resume)
