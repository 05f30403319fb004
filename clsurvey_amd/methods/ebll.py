"""EBLL (Encoder Based Lifelong Learning) on the HIP path — mirror of src/methods/EBLL/{AlexNet_EBLL,Finetune_SGD_EBLL}.py.

Two stages per task (method.py:822-936):
  1. `fine_tune_Adam_Autoencoder`: an under-complete autoencoder (Linear + Sigmoid code, Linear decoder) is inserted between
     the feature extractor and the classifier of the PREVIOUS task's model and trained with Adadelta on
         CrossEntropy(classifier(decode(encode(F(x))))) + alpha * MSE(decode(encode(F(x))), F(x))
     (Finetune_SGD_EBLL.py:99-199, 441-504); only the autoencoder's four tensors are optimised.
  2. `fine_tune_SGD_EBLL`: LwF's objective plus the code loss: the codes sigmoid(W_i F(x) + b_i) of every earlier task's
     encoder must stay where the frozen previous model puts them,
         total = CE(new head) + lambda * sum distillation(old heads) + reg_alpha * sum MSE(code_i, code_i of the old model)
     (Finetune_SGD_EBLL.py:219-395, 507-518).

Execution: the feature extractor / classifier run on the static-plan executor exactly as for LwF (stacked heads as ONE
Linear, `clhip_lwf_loss`); the code layers are a side branch off the flattened features (`NetEngine.layer_input`), their
loss gradient comes back through `clhip_sigmoid_bwd` / `clhip_fc_bwd_data` and is added to the feature gradient inside the
plan's backward (`clhip_net_set_input_grad`).  Stage 1 runs the (frozen) feature extractor forward-only on the executor
and the small autoencoder + classifier tail on the autograd bridges.
"""
import copy
import os
import time

import torch
from ..data import load_task_datasets
import torch.nn as nn

from .. import ops
from ..models import _walk
from ..net import NetEngine
from ..optim import SGD, Adadelta
from . import train_common as tc
from .lwf import LwfEngine


class AutoEncoder(nn.Module):
    """AlexNet_EBLL.py:9-26."""

    def __init__(self, x_dim, h1_dim):
        super().__init__()
        self.encode = nn.Sequential(nn.Linear(x_dim, h1_dim), nn.Sigmoid())
        self.decode = nn.Sequential(nn.Linear(h1_dim, x_dim))

    def forward(self, x):
        return _decode(self, _encode(self.encode, x))


def _encode(encode, x):
    lin = encode[0]
    return ops.sigmoid(ops.linear(x, lin.weight, lin.bias, False))


def _decode(autoencoder, h):
    lin = autoencoder.decode[0]
    return ops.linear(h, lin.weight, lin.bias, False)


class autoencoders(nn.Module):
    """AlexNet_EBLL.py:29-41: the encoders of all earlier tasks, applied to the same features."""

    def __init__(self, autoencoder):
        super().__init__()
        self.add_module("0", autoencoder.encode)

    def forward(self, x):
        return [_encode(module, x) for module in self._modules.values()]


def _classifier_tail(classifier, x, last_layer_name, training):
    """Modules 0 .. last_layer_name-1 in sequence, then every later module (a head) on the same input."""
    cls = list(classifier.children())
    x = _walk(x, cls[:last_layer_name], training, "EBLL classifier")
    return [ops.linear(x, h.weight, h.bias, False) for h in cls[last_layer_name:]]


class AlexNet_ENCODER(nn.Module):
    """AlexNet_EBLL.py:44-89: net with the autoencoder between feature extractor and classifier; forward returns
    (output of the LAST head, autoencoder input, autoencoder reconstruction)."""

    def __init__(self, alexnet, dim=100, last_layer_name=6, num_ftrs=256 * 6 * 6):
        super().__init__()
        self.add_module("features", alexnet.features)
        self.add_module("autoencoder", AutoEncoder(num_ftrs, dim))
        self.add_module("classifier", alexnet.classifier)
        self.last_layer_name = last_layer_name

    def forward(self, x):
        x = torch.flatten(_walk(x, list(self.features.children()), self.training, "EBLL features"), 1)
        encoder_input = x.detach().clone()
        recon = self.autoencoder(x)
        outs = _classifier_tail(self.classifier, recon, int(self.last_layer_name), self.training)
        return outs[-1], encoder_input, recon


class AlexNet_EBLL(nn.Module):
    """AlexNet_EBLL.py:92-138: forward returns (list of head outputs — or the single one in finetune mode —, codes)."""

    def __init__(self, model, autoencoder, last_layer_name=6):
        super().__init__()
        self.add_module("features", model.features)
        self.add_module("autoencoders", autoencoders(autoencoder))
        self.add_module("classifier", model.classifier)
        self.last_layer_name = last_layer_name
        self.finetune_mode = False

    def set_finetune_mode(self, mode):
        self.finetune_mode = mode

    def forward(self, x):
        x = torch.flatten(_walk(x, list(self.features.children()), self.training, "EBLL features"), 1)
        codes = self.autoencoders(x)
        outputs = _classifier_tail(self.classifier, x, int(self.last_layer_name), self.training)
        if self.finetune_mode:
            assert len(outputs) == 1
            outputs = outputs[0]
        return outputs, codes


# --------------------------------------------------------------------------------------------- stage 1: autoencoder
class _FeatureNet(nn.Module):
    """features-only view for the plan executor (same module objects)."""

    def __init__(self, features):
        super().__init__()
        self.features = features
        self.classifier = nn.Sequential()


def train_autoencoder(model, optimizer, lr, dset_loaders, dset_sizes, num_epochs, exp_dir="./", resume="", alpha=1e-6,
                      feature_engine=None):
    """Finetune_SGD_EBLL.py:99-199. Returns (model, best validation accuracy in [0, 1])."""
    best_acc, count, start_epoch = 0, 0, 0
    if os.path.isfile(resume):
        ck = torch.load(resume, weights_only=False)
        start_epoch, count, best_acc = ck["epoch"], ck["count"], ck["best_acc"]
        model.load_state_dict(ck["state_dict"])
        optimizer.load_state_dict(ck["optimizer"])
    frozen = [p for p in list(model.features.parameters()) + list(model.classifier.parameters())]
    flags = [p.requires_grad for p in frozen]
    stats = torch.zeros(3, dtype=torch.float64, device=feature_engine.device)     # total loss, encoder loss, hits
    try:
        for p in frozen:                  # only the autoencoder is optimised (:497): skip the unused weight gradients
            p.requires_grad_(False)
        for epoch in range(start_epoch, num_epochs):
            print("Epoch {}/{}".format(epoch, num_epochs - 1))
            for phase in ("train", "val"):
                model.train(phase == "train")
                feature_engine.model.train(phase == "train")
                stats.zero_()
                for inputs, labels in dset_loaders[phase]:
                    feat = feature_engine.forward(inputs)                      # F(x), flattened, no gradient
                    optimizer.zero_grad()
                    with torch.enable_grad():
                        recon = model.autoencoder(feat)
                        out = _classifier_tail(model.classifier, recon, int(model.last_layer_name), model.training)[-1]
                        task_loss = ops.cross_entropy(out, labels)
                        encoder_loss = ops.mse_loss(recon, feat)
                        total_loss = alpha * encoder_loss + task_loss
                    if phase == "train":
                        total_loss.backward()
                        optimizer.step()
                    with torch.no_grad():
                        stats[0] += total_loss.detach().double()
                        stats[1] += encoder_loss.detach().double()
                        stats[2] += (out.detach().argmax(1) == labels).sum().double()
                s = stats.cpu()
                epoch_loss, enc_loss, epoch_acc = float(s[0]) / dset_sizes[phase], float(s[1]) / dset_sizes[phase], \
                    float(s[2]) / dset_sizes[phase]
                print("{} Loss: {:.4f} Acc: {:.4f}".format(phase, epoch_loss, epoch_acc))
                print("{} Encoder Loss: {:.4f} Acc: {:.4f}".format(phase, enc_loss, epoch_acc))
                if phase == "val":
                    if epoch_acc > best_acc:
                        best_acc, count = epoch_acc, 0
                        _save_with_flags(model, frozen, flags, os.path.join(exp_dir, "best_model.pth.tar"))
                    else:
                        if count == 5:
                            return model, best_acc
                        count += 1
            _save_with_flags({"epoch": epoch + 1, "encoder_loss": enc_loss, "count": count, "best_acc": best_acc,
                              "arch": "alexnet", "model": model, "state_dict": model.state_dict(),
                              "optimizer": optimizer.state_dict()}, frozen, flags, os.path.join(exp_dir, "epoch.pth.tar"))
    finally:
        for p, f in zip(frozen, flags):
            p.requires_grad_(f)
    return model, best_acc


def _save_with_flags(obj, frozen, flags, path):
    """Pickle with the parameters' own requires_grad flags (they are switched off only while stage 1 runs)."""
    for p, f in zip(frozen, flags):
        p.requires_grad_(f)
    try:
        torch.save(obj, path)
    finally:
        for p in frozen:
            p.requires_grad_(False)


def get_first_FC_layer(seq_module):
    """utilities/utils.py: first nn.Linear of a Sequential."""
    for module in seq_module.modules():
        if isinstance(module, nn.Linear):
            return module
    raise Exception("No LINEAR module in sequential found...")


def fine_tune_Adam_Autoencoder(dataset_path, previous_task_model_path, exp_dir="", batch_size=200, num_epochs=100, lr=0.01,
                               pretrained=True, alpha=1e-6, auto_dim=100, last_layer_name=6, device="cuda"):
    """Finetune_SGD_EBLL.py:441-504. Returns (AlexNet_ENCODER, best validation accuracy)."""
    dsets = load_task_datasets(dataset_path)
    dset_loaders = tc.make_loaders(dsets, batch_size, device)
    dset_sizes = {x: len(dsets[x]) for x in ["train", "val"]}
    resume = os.path.join(exp_dir, "epoch.pth.tar")
    if os.path.isfile(resume):
        model_ft = torch.load(resume, weights_only=False)["model"]
    else:
        model_ft = tc.load_model(previous_task_model_path)
        num_ftrs = get_first_FC_layer(model_ft.classifier).in_features
        if hasattr(model_ft, "reg_params"):
            model_ft.reg_params = None
        model_ft = AlexNet_ENCODER(model_ft, dim=auto_dim, last_layer_name=last_layer_name, num_ftrs=num_ftrs)
    os.makedirs(exp_dir, exist_ok=True)
    model_ft = model_ft.to(device)
    in_shape = tuple(dset_loaders["train"].x.shape[1:])
    feature_engine = NetEngine(_FeatureNet(model_ft.features), batch_size, in_shape, device)
    optimizer_ft = Adadelta(model_ft.autoencoder.parameters(), lr)
    return train_autoencoder(model_ft, optimizer_ft, lr, dset_loaders, dset_sizes, num_epochs, exp_dir, resume, alpha=alpha,
                             feature_engine=feature_engine)


# --------------------------------------------------------------------------------------------- stage 2: EBLL training
class EbllEngine:
    """LwfEngine + the code layers as a side branch off the flattened features."""

    def __init__(self, wrapper, max_batch, in_shape, device="cuda"):
        self.lwf = LwfEngine(wrapper, max_batch, in_shape, device)
        eng = self.lwf.engine
        self.fc_first = next(i for i, sp in enumerate(eng.layers) if sp[0] == "fc")
        self.wrapper = wrapper
        self.device = eng.device
        self.code_loss = torch.zeros(1, dtype=torch.float32, device=self.device)

    @property
    def arena(self):
        return self.lwf.arena

    def features(self, n):
        return self.lwf.engine.layer_input(self.fc_first, n)

    def codes(self, feat):
        out = []
        for enc in self.wrapper.autoencoders._modules.values():
            lin = enc[0]
            out.append(ops.sigmoid_fwd(ops.fc_fwd(feat, lin.weight.data, lin.bias.data, False)))
        return out

    def targets(self, x):
        """Frozen previous model: (logits of its heads, its codes)."""
        logits = self.lwf.logits(x)
        return logits, self.codes(self.features(x.shape[0]))

    def step(self, x, y, target_logits, target_codes, T, reg_lambda, reg_alpha, backward=True, stats=None):
        """One batch; returns (loss2 [task CE, lambda * distillation], code loss [1]) on the device."""
        eng = self.lwf.engine
        if not backward:
            return self.lwf.step(x, y, None, T, reg_lambda, backward=False, stats=stats), None
        n = x.shape[0]
        z = eng.forward(x)
        feat = self.features(n)
        self.code_loss.zero_()
        extra = None
        for enc, tgt in zip(self.wrapper.autoencoders._modules.values(), target_codes):
            lin = enc[0]
            code = ops.sigmoid_fwd(ops.fc_fwd(feat, lin.weight.data, lin.bias.data, False))
            loss, dcode = ops.mse_mean(code, tgt, grad_scale=reg_alpha)
            self.code_loss += loss
            dfeat = ops.fc_bwd_data(ops.sigmoid_bwd(dcode, code), lin.weight.data)
            extra = dfeat if extra is None else extra.add_(dfeat)
        if extra is not None:
            extra = ops.relu_bwd(extra, feat)            # the features are a (pooled) ReLU output
        eng.set_input_grad(self.fc_first, extra)
        try:
            loss2 = self.lwf.step(x, y, target_logits, T, reg_lambda, backward=True, stats=stats, z=z)
        finally:
            eng.set_input_grad(self.fc_first, None)
        return loss2, self.code_loss


def train_model_ebll(model, original_model, optimizer, lr, dset_loaders, dset_sizes, num_epochs, exp_dir="./", resume="",
                     temperature=2, reg_alpha=1e-6, saving_freq=5, reg_lambda=1, engine=None, teacher=None):
    """Finetune_SGD_EBLL.py:219-395. Returns (model, best validation accuracy in [0, 1])."""
    since = time.time()
    preprocessing_time = 0.0
    val_beat_counts, best_acc, start_epoch = 0, 0.0, 0
    if os.path.isfile(resume):
        ck = torch.load(resume, weights_only=False)
        start_epoch, best_acc, lr, val_beat_counts = ck["epoch"], ck["best_acc"], ck["lr"], ck["val_beat_counts"]
        model.load_state_dict(ck["state_dict"])
        optimizer.load_state_dict(ck["optimizer"])
    stats = torch.zeros(2, dtype=torch.float64, device=engine.device)
    code_sum = torch.zeros(1, dtype=torch.float64, device=engine.device)
    for epoch in range(start_epoch, num_epochs):
        print("Epoch {}/{}".format(epoch, num_epochs - 1))
        for phase in ("train", "val"):
            if phase == "train":
                optimizer, lr, cont = tc.set_lr(optimizer, lr, val_beat_counts)
                if not cont:
                    tc.save_preprocessing_time(exp_dir, preprocessing_time)
                    return model, best_acc
            model.train(phase == "train")
            stats.zero_()
            code_sum.zero_()
            for inputs, labels in dset_loaders[phase]:
                t0 = time.time()
                if phase == "train":
                    target_logits, target_codes = teacher.targets(inputs)
                    _, code_loss = engine.step(inputs, labels, target_logits, target_codes, temperature, reg_lambda, reg_alpha,
                                               backward=True, stats=stats)
                    preprocessing_time += time.time() - t0
                    optimizer.step()
                    code_sum += code_loss.double()
                else:
                    engine.step(inputs, labels, None, None, temperature, reg_lambda, reg_alpha, backward=False, stats=stats)
            s = stats.cpu()
            epoch_loss, epoch_acc = float(s[0]) / dset_sizes[phase], float(s[1]) / dset_sizes[phase]
            print("{} Loss: {:.4f} Acc: {:.4f}".format(phase, epoch_loss, epoch_acc))
            if phase == "train":
                print("{} code Loss: {:.4f}".format(phase, float(code_sum) / dset_sizes[phase]))
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


def fine_tune_SGD_EBLL(dataset_path, previous_task_model_path, autoencoder_model_path, init_model_path="", exp_dir="",
                       batch_size=200, num_epochs=100, lr=0.0004, init_freeze=1, weight_decay=0, reg_alpha=1e-6, saving_freq=5,
                       reg_lambda=1, device="cuda"):
    """Finetune_SGD_EBLL.py:507-518 + the model surgery above it. Returns train_model_ebll's (model, acc)."""
    dsets = load_task_datasets(dataset_path)
    dset_loaders = tc.make_loaders(dsets, batch_size, device)
    dset_sizes = {x: len(dsets[x]) for x in ["train", "val"]}
    dset_classes = dsets["train"].classes
    autoencoder_model = torch.load(autoencoder_model_path, weights_only=False)
    resume = os.path.join(exp_dir, "epoch.pth.tar")
    model_ft = tc.load_model(previous_task_model_path)
    if not type(model_ft) is AlexNet_EBLL:
        last_layer_index = len(model_ft.classifier._modules) - 1
        model_ft = AlexNet_EBLL(model_ft, autoencoder_model.autoencoder, last_layer_name=last_layer_index)
    else:
        model_ft.autoencoders.add_module(str(len(model_ft.autoencoders._modules.items())), autoencoder_model.autoencoder.encode)
    if hasattr(model_ft, "reg_params") and isinstance(model_ft.reg_params, dict):
        model_ft.reg_params.pop("__arena__", None)
    original_model = copy.deepcopy(model_ft)
    num_ftrs = model_ft.classifier[model_ft.last_layer_name].in_features
    n_mod = str(len(model_ft.classifier._modules))
    if not init_freeze:
        model_ft.classifier.add_module(n_mod, nn.Linear(num_ftrs, len(dset_classes)))
    else:
        init_model = tc.load_model(init_model_path)
        model_ft.classifier.add_module(n_mod, init_model.classifier[model_ft.last_layer_name])
        del init_model
    os.makedirs(exp_dir, exist_ok=True)
    model_ft, original_model = model_ft.to(device), original_model.to(device)
    model_ft.reg_params = {"lambda": reg_lambda, "reg_alpha": reg_alpha}
    in_shape = tuple(dset_loaders["train"].x.shape[1:])
    engine = EbllEngine(model_ft, batch_size, in_shape, device)
    teacher = EbllEngine(original_model, batch_size, in_shape, device)
    original_model.eval()
    optimizer_ft = SGD(engine.arena.params, lr, momentum=0.9, weight_decay=weight_decay)     # features + classifier (:513-514)
    return train_model_ebll(model_ft, original_model, optimizer_ft, lr, dset_loaders, dset_sizes, num_epochs, exp_dir, resume,
                            reg_alpha=reg_alpha, saving_freq=saving_freq, reg_lambda=reg_lambda, engine=engine, teacher=teacher)
