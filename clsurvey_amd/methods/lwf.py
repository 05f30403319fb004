"""LwF on the HIP path — mirror of src/methods/LwF/{AlexNet_LwF,main_LWF}.py.

The reference stacks one Linear head per task at the end of `model.classifier` (AlexNet_LwF.py:16-27 applies every
module from index `last_layer_name` on to the same shared features) and trains with
    total = CrossEntropy(new head) + lambda * sum_old distillation(old head, frozen previous model)   (main_LWF.py:184-202).
Here the stacked heads run as ONE Linear: their weights (and biases) are laid out back to back in the parameter arena,
the plan executor sees a [sum of head sizes]-way layer, and clhip_lwf_loss turns the side-by-side logits + the teacher's
logits into the loss and d loss / d logits in one launch.  The teacher forward is one plan execution on its own arena.
"""
import copy
import ctypes as C
import os
import time

import torch
from ..data import load_task_datasets
import torch.nn as nn

from .. import _lib, ops
from .._lib import check
from ..net import NetEngine
from ..optim import SGD
from . import train_common as tc


class AlexNet_LwF(nn.Module):
    """AlexNet_LwF.py:4-38 — same attributes and forward contract (list of head outputs, or the single output in
    finetune mode); forward runs on the HIP autograd bridges (evaluation / get_output path)."""

    def __init__(self, model, last_layer_name=6):
        super().__init__()
        self.model = model
        self.last_layer_name = last_layer_name
        self.finetune_mode = False

    def set_finetune_mode(self, mode):
        self.finetune_mode = mode

    def forward(self, x):
        from ..models import _walk
        x = _walk(x, list(self.model.features.children()), self.training, "LwF features")
        x = torch.flatten(x, 1)
        cls = list(self.model.classifier.children())
        x = _walk(x, cls[:self.last_layer_name], self.training, "LwF classifier")
        outputs = [ops.linear(x, h.weight, h.bias, False) for h in cls[self.last_layer_name:]]
        if self.finetune_mode:
            assert len(outputs) == 1
            outputs = outputs[0]
        return outputs


def lwf_plan(wrapper):
    """(layers, params, head_sizes) for NetEngine: conv / shared Linear layers as they are, all heads as ONE Linear whose
    rows are the heads' rows back to back (arena order: ..., head weights, head biases)."""
    from ..net import parse_net, conv_geometry
    m = getattr(wrapper, "model", wrapper)          # AlexNet_LwF holds the net in .model, AlexNet_EBLL is the net
    cls = list(m.classifier.children())
    shared = nn.Module()
    shared.features = m.features
    shared.classifier = nn.Sequential(*cls[:wrapper.last_layer_name])
    base, drops = parse_net(shared)
    layers = [(kind, mod.weight, mod.bias, mod.in_channels if kind == "conv" else mod.in_features,
               mod.out_channels if kind == "conv" else mod.out_features, relu, pool) +
              ((conv_geometry(mod),) if kind == "conv" else ()) for kind, mod, relu, pool in base]
    heads = cls[wrapper.last_layer_name:]
    if not heads or any(not isinstance(h, nn.Linear) for h in heads):
        raise NotImplementedError("LwF: the modules from last_layer_name on must be Linear heads")
    feat = heads[0].in_features
    for h in heads:
        if (h.out_features * feat) % 4 or h.out_features % 4:
            raise NotImplementedError("LwF head sizes must be multiples of 4 (arena slots are 16-byte aligned)")
    sizes = [h.out_features for h in heads]
    layers.append(("fc", heads[0].weight, heads[0].bias, feat, sum(sizes), False, False))
    params = [p for _, w, b, *_ in layers[:-1] for p in (w, b)] + [h.weight for h in heads] + [h.bias for h in heads]
    return layers, params, sizes, drops


class LwfEngine:
    """forward of all heads + LwF loss + backward for one batch."""

    def __init__(self, wrapper, max_batch, in_shape, device="cuda"):
        layers, params, self.sizes, drops = lwf_plan(wrapper)
        self.engine = NetEngine(wrapper, max_batch, in_shape, device, layers=layers, params=params, drops=drops)
        self.device = self.engine.device
        self.n_out = sum(self.sizes)
        self._sizes = (C.c_int * len(self.sizes))(*self.sizes)
        self.loss2 = torch.zeros(2, dtype=torch.float32, device=self.device)
        self.dlogits = torch.zeros((max_batch, self.n_out), dtype=torch.float32, device=self.device)

    @property
    def arena(self):
        return self.engine.arena

    def logits(self, x):
        return self.engine.forward(x)

    def step(self, x, y, teacher_logits, T, reg_lambda, backward=True, stats=None, z=None):
        """Returns loss2 (device: [task CE, lambda * sum distillation]). teacher_logits: [N][sum of the OLD head sizes];
        z: the logits of a forward the caller has just run on this engine for the same x (else it runs here)."""
        if z is None:
            z = self.engine.forward(x)
        n = x.shape[0]
        dl = self.dlogits[:n]
        check(_lib.lib().clhip_lwf_loss(
            z.data_ptr(), y.data_ptr(), teacher_logits.data_ptr() if teacher_logits is not None else None, self._sizes,
            len(self.sizes), n, self.n_out, teacher_logits.shape[1] if teacher_logits is not None else 0, float(T),
            float(reg_lambda), int(backward and teacher_logits is not None), dl.data_ptr(), self.loss2.data_ptr(),
            stats.data_ptr() if stats is not None else None, torch.cuda.current_stream().cuda_stream), "clhip_lwf_loss")
        if backward:
            self.engine.backward(x, dl)
        return self.loss2


def train_model_lwf(model, original_model, optimizer, lr, dset_loaders, dset_sizes, num_epochs, exp_dir="./", resume="",
                    temperature=2, saving_freq=5, reg_lambda=1, engine=None, teacher=None):
    """main_LWF.py:100-250. Returns (model, best validation accuracy in [0, 1])."""
    since = time.time()
    val_beat_counts, best_acc, start_epoch = 0, 0.0, 0
    if os.path.isfile(resume):
        ck = torch.load(resume, weights_only=False)
        start_epoch, best_acc, lr, val_beat_counts = ck["epoch"], ck["best_acc"], ck["lr"], ck["val_beat_counts"]
        model.load_state_dict(ck["state_dict"])
        optimizer.load_state_dict(ck["optimizer"])
    stats = torch.zeros(2, dtype=torch.float64, device=engine.device)
    preprocessing_time = 0.0
    if original_model is not None:
        original_model.eval()                   # main_LWF.py:104: the frozen teacher never drops units
    for epoch in range(start_epoch, num_epochs):
        print("Epoch {}/{}".format(epoch, num_epochs - 1))
        for phase in ("train", "val"):
            if phase == "train":
                optimizer, lr, cont = tc.set_lr(optimizer, lr, val_beat_counts)
                if not cont:
                    tc.save_preprocessing_time(exp_dir, preprocessing_time)
                    return model, best_acc
            model.train(phase == "train")       # main_LWF.py:145-147
            stats.zero_()
            for inputs, labels in dset_loaders[phase]:
                t0 = time.time()
                if phase == "train":
                    target = teacher.logits(inputs)                     # frozen previous model, all old heads
                    engine.step(inputs, labels, target, temperature, reg_lambda, backward=True, stats=stats)
                    preprocessing_time += time.time() - t0
                    optimizer.step()
                else:
                    engine.step(inputs, labels, None, temperature, reg_lambda, backward=False, stats=stats)
            s = stats.cpu()
            epoch_loss, epoch_acc = float(s[0]) / dset_sizes[phase], float(s[1]) / dset_sizes[phase]
            print("{} Loss: {:.4f} Acc: {:.4f}".format(phase, epoch_loss, epoch_acc))
            if phase == "val":
                if epoch_acc > best_acc:
                    best_acc = epoch_acc
                    tc.save_model(model, os.path.join(exp_dir, "best_model.pth.tar"))
                    val_beat_counts = 0
                else:
                    val_beat_counts += 1
        if epoch % saving_freq == 0:
            torch.save({"epoch": epoch + 1, "lr": lr, "val_beat_counts": val_beat_counts, "epoch_acc": epoch_acc,
                        "best_acc": best_acc, "arch": "alexnet", "model": model, "state_dict": model.state_dict(),
                        "optimizer": optimizer.state_dict()}, os.path.join(exp_dir, "epoch.pth.tar"))
    print("Training complete in {:.0f}s, best val acc {:.4f}".format(time.time() - since, best_acc))
    tc.save_preprocessing_time(exp_dir, preprocessing_time)
    return model, best_acc


def fine_tune_freeze(dataset_path, model_path, exp_dir, batch_size=100, num_epochs=100, lr=0.0004, device="cuda"):
    """main_LWF.py:322-362 — LwF's optional head warm-up: a fresh head for the new task on top of the previous model,
    trained alone (plain CE, SGD momentum 0.9 over the head's two tensors) with everything else frozen; best_model.pth.tar
    and epoch.pth.tar land in exp_dir.  A wrapper from an earlier LwF task is cut back to its net with the shared layers
    + one head slot (the reference reaches for `.module` there, an attribute AlexNet_LwF does not have, and stops)."""
    print("lr is " + str(lr))
    dsets = load_task_datasets(dataset_path)
    dset_loaders = tc.make_loaders(dsets, batch_size, device)
    dset_sizes = {x: len(dsets[x]) for x in ["train", "val"]}
    model_ft = tc.load_model(model_path)
    if isinstance(model_ft, AlexNet_LwF):
        net = model_ft.model
        net.classifier = nn.Sequential(*list(net.classifier.children())[:model_ft.last_layer_name + 1])
        model_ft = net
    tc.replace_head(model_ft, len(dsets["train"].classes))
    os.makedirs(exp_dir, exist_ok=True)
    model_ft = model_ft.to(device)
    engine = tc.engine_for(model_ft, dset_loaders, batch_size, device)
    last = str(len(model_ft.classifier._modules) - 1)
    optimizer_ft = SGD(model_ft.classifier._modules[last].parameters(), lr, momentum=0.9)
    model_ft, _ = tc.train_model(model_ft, engine, optimizer_ft, lr, dset_loaders, dset_sizes, num_epochs, exp_dir,
                                 os.path.join(exp_dir, "epoch.pth.tar"), step_fn=optimizer_ft.step, abort_on_bad_loss=False)
    return model_ft


def fine_tune_SGD_LwF(dataset_path, previous_task_model_path, init_model_path="", exp_dir="", batch_size=200,
                      num_epochs=100, lr=0.0004, init_freeze=1, pretrained=True, weight_decay=0, last_layer_name=6,
                      saving_freq=5, reg_lambda=1, device="cuda"):
    """main_LWF.py:253-318. Returns what the reference returns: train_model_lwf's (model, acc) tuple."""
    dsets = load_task_datasets(dataset_path)
    dset_loaders = tc.make_loaders(dsets, batch_size, device)
    dset_sizes = {x: len(dsets[x]) for x in ["train", "val"]}
    dset_classes = dsets["train"].classes
    resume = os.path.join(exp_dir, "epoch.pth.tar")
    if os.path.isfile(resume):
        model_ft = torch.load(resume, weights_only=False)["model"]
        previous_model = tc.load_model(previous_task_model_path)
        if not isinstance(previous_model, AlexNet_LwF):
            previous_model = AlexNet_LwF(previous_model, last_layer_name=last_layer_name)
        original_model = copy.deepcopy(previous_model)
    else:
        model_ft = tc.load_model(previous_task_model_path)
        if not isinstance(model_ft, AlexNet_LwF):
            last_layer_index = len(model_ft.classifier._modules) - 1
            model_ft = AlexNet_LwF(model_ft, last_layer_name=last_layer_index)
            model_ft.num_ftrs = model_ft.model.classifier[last_layer_index].in_features
        original_model = copy.deepcopy(model_ft)
        n_mod = str(len(model_ft.model.classifier._modules))
        if not init_freeze:
            model_ft.model.classifier.add_module(n_mod, nn.Linear(model_ft.num_ftrs, len(dset_classes)))
        else:
            init_model = tc.load_model(init_model_path)
            model_ft.model.classifier.add_module(n_mod, init_model.classifier[6])
        os.makedirs(exp_dir, exist_ok=True)
    if not hasattr(model_ft, "reg_params"):
        model_ft.reg_params = {}
    model_ft.reg_params["reg_lambda"] = reg_lambda
    model_ft, original_model = model_ft.to(device), original_model.to(device)
    in_shape = tuple(dset_loaders["train"].x.shape[1:])
    engine = LwfEngine(model_ft, batch_size, in_shape, device)
    teacher = LwfEngine(original_model, batch_size, in_shape, device)
    optimizer_ft = SGD(engine.arena.params, lr, momentum=0.9, weight_decay=weight_decay)
    return train_model_lwf(model_ft, original_model, optimizer_ft, lr, dset_loaders, dset_sizes, num_epochs, exp_dir, resume,
                           saving_freq=saving_freq, reg_lambda=reg_lambda, engine=engine, teacher=teacher)
