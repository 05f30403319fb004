"""IMM on the HIP path — mirror of src/methods/IMM/{main_L2transfer,train_L2transfer,merge}.py.

Training = L2-transfer: the EWC optimizer with Omega = 1 and theta* = the previous task's weights
(train_L2transfer.py:20-100 is Weight_Regularized_SGD again; note that update_reg_params runs AFTER the head is
replaced, so the fresh head is pulled towards its own random initialisation, main_L2transfer.py:118-139).
After training, models are merged per task (merge.py): mean-IMM averages the task models, mode-IMM weights
them by their diagonal Fisher estimated with labels SAMPLED from the model's own softmax (merge.py:155-183).
"""
import copy
import ctypes as C
import os
import time

import torch
from ..data import load_task_datasets

from .. import _lib, ops
from .._lib import check
from ..data import DeviceLoader
from ..net import NetEngine
from ..optim import Weight_Regularized_SGD, arena_reg_params
from . import train_common as tc


def update_reg_params(model, freeze_layers=None):
    """main_L2transfer.py:29-70: omega = ones, init_val = theta for every parameter (frozen ones dropped)."""
    reg_params = model.reg_params
    freeze_layers = [] if freeze_layers is None else freeze_layers
    for name, param in model.named_parameters():
        if param in reg_params and name in freeze_layers:
            del reg_params[param]
        else:
            reg_params[param] = {"omega": torch.ones_like(param.data), "init_val": param.data.clone()}
    return reg_params


def fine_tune_l2transfer(dataset_path, model_path, exp_dir, batch_size=100, num_epochs=100, lr=0.0004, reg_lambda=100,
                         init_freeze=0, weight_decay=0, saving_freq=5, device="cuda"):
    """main_L2transfer.py:73-158."""
    dsets = load_task_datasets(dataset_path)
    dset_loaders = tc.make_loaders(dsets, batch_size, device)
    dset_sizes = {x: len(dsets[x]) for x in ["train", "val"]}
    dset_classes = dsets["train"].classes
    resume = os.path.join(exp_dir, "epoch.pth.tar")
    if os.path.isfile(resume):
        model_ft = torch.load(resume, weights_only=False)["model"]
    else:
        model_ft = tc.load_model(model_path)
    if not init_freeze:
        tc.replace_head(model_ft, len(dset_classes))
    os.makedirs(exp_dir, exist_ok=True)
    model_ft = model_ft.to(device)
    if not os.path.isfile(resume):
        if not hasattr(model_ft, "reg_params"):
            model_ft.reg_params = {}
        parameters = list(model_ft.parameters())
        model_ft.reg_params.pop(parameters[-1], None)
        model_ft.reg_params.pop(parameters[-2], None)
        model_ft.reg_params.pop("__arena__", None)
        reg_params = update_reg_params(model_ft)
        reg_params["lambda"] = reg_lambda
        model_ft.reg_params = reg_params
    engine = tc.engine_for(model_ft, dset_loaders, batch_size, device)
    arena_reg_params(engine.arena, model_ft.reg_params)
    optimizer_ft = Weight_Regularized_SGD(model_ft.parameters(), lr, momentum=0.9, weight_decay=weight_decay)
    return tc.train_model(model_ft, engine, optimizer_ft, lr, dset_loaders, dset_sizes, num_epochs, exp_dir, resume,
                          saving_freq=saving_freq, abort_on_bad_loss=False)


def sample_targets(logits):
    """merge.py:172-174: torch.multinomial(softmax(output), 1).squeeze() (device RNG)."""
    return torch.multinomial(torch.softmax(logits, dim=1), 1).squeeze(1)


def diag_fisher(model, dataset, exclude_params=None, sampler=sample_targets, engine=None):
    """merge.py:155-183: precision[n] = 1e-8 + sum over BOTH phases of grad(mean nll(sampled labels))^2 / #batches.
    `dataset` = {'train': loader, 'val': loader}. Per batch: forward (logits), sample labels, one fused
    forward + CE(mean) + backward, then clhip_fisher_accum over the whole arena."""
    exclude_params = exclude_params or []
    model.eval()                                # merge.py:165
    loaders = list(dataset.values())
    if engine is None:
        engine = NetEngine(model, loaders[0].batch_size, tuple(loaders[0].x.shape[1:]), loaders[0].device)
    A = engine.arena
    prec = A.buffer("imm_precision")
    prec.fill_(1e-8)
    for loader in loaders:
        for x, _ in loader:
            targets = sampler(engine.forward(x))
            engine.loss_step(x, targets.contiguous(), "ce_mean", backward=True)
            ops.fisher_accum(prec, A.grad, float(len(loader)))       # len(dataset[phase]) = number of batches
    return {n: A.view("imm_precision", p).clone() for n, p in model.named_parameters() if n not in exclude_params}


def _merge_tensor(thetas, precisions, sum_precision, out):
    L = _lib.lib()
    m = len(thetas)
    tp = (C.c_void_p * m)(*[t.data_ptr() for t in thetas])
    pp = (C.c_void_p * m)(*[p.data_ptr() for p in precisions]) if precisions is not None else None
    check(L.clhip_imm_merge(tp, pp, sum_precision.data_ptr() if sum_precision is not None else None, m, out.numel(),
                            out.data_ptr(), torch.cuda.current_stream().cuda_stream), "clhip_imm_merge")


def IMM_merge_models(models, task_list_idx, head_param_names, precision=None, sum_precision=None, mean_mode=True,
                     device="cuda", fix_mean=False):
    """merge.py:185-242. fix_mean=True applies the average the reference meant to compute in mean mode."""
    if not mean_mode and (precision is None or sum_precision is None):
        raise Exception("Can only use precision for MODE IMM, not mean IMM")
    merged_model = copy.deepcopy(models[task_list_idx]).to(device)
    total = task_list_idx + 1
    states = [m.state_dict() for m in models[:total]]
    for name, param in merged_model.named_parameters():
        if name in head_param_names:
            continue
        thetas = [s[name].to(device, torch.float32).contiguous() for s in states]
        for t in thetas:
            if t.shape != param.shape:
                raise Exception("ERROR WHEN MERGING MODELS: PRECEDING MODEL PARAMS != PARAM SIZE OF REF TASK " + str(task_list_idx))
        out = torch.empty_like(thetas[0])
        if mean_mode:
            if not fix_mean:
                # Reference behaviour (pinned by tests/golden/G13): merge.py:223-224 rebinds the loop variable
                # `param_value` to a preceding model's state_dict tensor, so `param_value.data = mean_param` (:239)
                # never reaches the merged model — mean-IMM evaluates the task's own, unmerged weights.
                continue
            _merge_tensor(thetas, None, None, out)
        else:
            precs = [precision[i][name].to(device, torch.float32).contiguous() for i in range(total)]
            _merge_tensor(thetas, precs, sum_precision[name].to(device, torch.float32).contiguous(), out)
        param.data = out
    return merged_model


def preprocess_merge_IMM(method, model_paths, datasets_path, batch_size, overwrite=False, device="cuda"):
    """merge.py:12-150. Returns the list of model paths to evaluate (first task's model + merged models)."""
    mode = method.mode
    merge_name = "best_model_" + mode + "_merge.pth.tar"
    models = [tc.load_model(p) for p in model_paths]
    merged_paths = [model_paths[0]]
    last = str(len(models[0].classifier._modules) - 1)
    head_param_names = ["classifier.{}.{}".format(last, n) for n, _ in models[0].classifier._modules[last].named_parameters()]
    precision_matrices, sum_precision_matrices = [], []
    if mode == method.modes[1]:
        t0 = time.time()
        sum_precision = None
        for i, model in enumerate(models):
            out_file = os.path.join(os.path.dirname(model_paths[i]), "precision_" + mode + ".pth.tar")
            if os.path.exists(out_file) and not overwrite:
                prec = torch.load(out_file, weights_only=False)
            else:
                dsets = load_task_datasets(datasets_path[i])
                loaders = {x: DeviceLoader(dsets[x], batch_size, True, device) for x in ["train", "val"]}
                prec = diag_fisher(model.to(device), loaders, exclude_params=head_param_names)
                torch.save(prec, out_file)
            precision_matrices.append(prec)
            if sum_precision is None:
                sum_precision = prec
            else:
                sum_precision = {n: p + prec[n] for n, p in sum_precision.items()}
                torch.save(sum_precision, os.path.join(os.path.dirname(model_paths[i]), "sum_precision_" + mode + ".pth.tar"))
                sum_precision_matrices.append(sum_precision)
        print("MODE IMM IWS: {:.1f}s".format(time.time() - t0))
    for i in range(1, len(models)):
        out_file = os.path.join(os.path.dirname(model_paths[i]), merge_name)
        if mode == method.modes[0]:
            merged = IMM_merge_models(models, i, head_param_names, mean_mode=True, device=device)
        else:
            merged = IMM_merge_models(models, i, head_param_names, precision=precision_matrices,
                                      sum_precision=sum_precision_matrices[i - 1], mean_mode=False, device=device)
        tc.save_model(merged, out_file)
        merged_paths.append(out_file)
    return merged_paths
