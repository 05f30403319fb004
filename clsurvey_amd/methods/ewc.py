"""EWC on the HIP path — mirror of src/methods/EWC/{main_EWC,train_EWC}.py.

Same function names / keyword signatures as the reference's L2 entry points:
fine_tune_EWC_acuumelation, accumulate_EWC_weights, diag_fisher, initialize_reg_params,
store_prev_reg_params, accumelate_reg_params.
"""
import os
import time

import torch
from ..data import load_task_datasets

from .. import ops
from ..data import DeviceLoader
from ..net import NetEngine
from ..optim import Weight_Regularized_SGD, arena_reg_params
from . import train_common as tc


def initialize_reg_params(model, freeze_layers=None):
    """main_EWC.py:160-174."""
    freeze_layers = freeze_layers or []
    reg_params = {}
    for name, param in model.named_parameters():
        if name not in freeze_layers:
            reg_params[param] = {"omega": torch.zeros_like(param.data), "init_val": param.data.clone()}
    return reg_params


def store_prev_reg_params(model, freeze_layers=None):
    """main_EWC.py:178-202: prev_omega <- omega, omega <- 0, init_val <- theta."""
    freeze_layers = freeze_layers or []
    reg_params = model.reg_params
    for name, param in model.named_parameters():
        if name not in freeze_layers:
            if param in reg_params:
                rp = reg_params[param]
                rp["prev_omega"] = rp["omega"]
                rp["omega"] = torch.zeros_like(param.data)
                rp["init_val"] = param.data.clone()
        elif param in reg_params:
            del reg_params[param]
    return reg_params


def accumelate_reg_params(model, freeze_layers=None):
    """main_EWC.py:205-232: omega <- prev_omega + omega."""
    freeze_layers = freeze_layers or []
    reg_params = model.reg_params
    for name, param in model.named_parameters():
        if name not in freeze_layers:
            if param in reg_params:
                rp = reg_params[param]
                rp["omega"] = torch.add(rp["prev_omega"].to(param.device), rp["omega"].to(param.device))
                del rp["prev_omega"]
        elif param in reg_params:
            del reg_params[param]
    return reg_params


def diag_fisher(model, dset_loader, data_len, engine=None):
    """main_EWC.py:138-157. One fused pass per batch: forward + nll(sum) + backward through the
    plan executor, then omega += grad^2 / data_len over the whole arena (clhip_fisher_accum).
    Parameters absent from reg_params keep no omega (their arena slot is scratch)."""
    reg_params = model.reg_params
    model.eval()                                # main_EWC.py:140: no Dropout noise, BatchNorm running statistics
    if engine is None:
        engine = NetEngine(model, dset_loader.batch_size, tuple(dset_loader.x.shape[1:]), dset_loader.device)
    A = engine.arena
    A.load("omega", {p: reg_params[p]["omega"] for p in A.params if p in reg_params})
    omega = A.aux["omega"]
    for x, label in dset_loader:
        engine.loss_step(x, label, "ce_sum", backward=True)
        ops.fisher_accum(omega, A.grad, float(data_len))
    for p in A.params:
        if p in reg_params:
            reg_params[p]["omega"] = A.view("omega", p).clone()
    return model


def accumulate_EWC_weights(data_dir, reg_sets, model_ft, batch_size, device="cuda", cache=None):
    """main_EWC.py:79-123 (data_dir is None on this path: reg_sets are pickled dataset dicts)."""
    dset_loader = None
    for data_path in reg_sets:
        dset = load_task_datasets(data_path)
        dset = dset["train"]
        dset_loader = DeviceLoader(dset, batch_size, False, device)
    if not hasattr(model_ft, "reg_params"):
        model_ft.reg_params = initialize_reg_params(model_ft)
    model_ft.reg_params = store_prev_reg_params(model_ft)
    data_len = len(dset)
    model_ft = diag_fisher(model_ft, dset_loader, data_len)
    model_ft.reg_params = accumelate_reg_params(model_ft)
    return model_ft


def fine_tune_EWC_acuumelation(dataset_path, previous_task_model_path, exp_dir, data_dir, reg_sets, reg_lambda=1,
                               num_epochs=100, lr=0.0008, batch_size=200, weight_decay=0, head_shared=False,
                               saving_freq=5, device="cuda"):
    """main_EWC.py:14-76."""
    dsets = load_task_datasets(dataset_path)
    dset_loaders = tc.make_loaders(dsets, batch_size, device)
    dset_sizes = {x: len(dsets[x]) for x in ["train", "val"]}
    dset_classes = dsets["train"].classes

    t0 = time.time()
    model_ft = tc.load_model(previous_task_model_path)
    model_ft = model_ft.to(device)
    model_ft = accumulate_EWC_weights(data_dir, reg_sets, model_ft, batch_size, device)
    model_ft.reg_params["lambda"] = reg_lambda
    tc.save_preprocessing_time(exp_dir, time.time() - t0)

    if not head_shared:
        tc.replace_head(model_ft, len(dset_classes))
    model_ft = model_ft.to(device)
    engine = tc.engine_for(model_ft, dset_loaders, batch_size, device)
    arena_reg_params(engine.arena, model_ft.reg_params)
    optimizer_ft = Weight_Regularized_SGD(model_ft.parameters(), lr, momentum=0.9, weight_decay=weight_decay)
    os.makedirs(exp_dir, exist_ok=True)
    resume = os.path.join(exp_dir, "epoch.pth.tar")
    model_ft, acc = tc.train_model(model_ft, engine, optimizer_ft, lr, dset_loaders, dset_sizes, num_epochs, exp_dir,
                                   resume, saving_freq=saving_freq)
    return model_ft, acc
