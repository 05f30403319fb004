"""SI on the HIP path — mirror of src/methods/SI/{main_SI,train_SI}.py."""
import os
import time

import torch
from ..data import load_task_datasets

from .. import ops
from ..optim import Elastic_SGD, arena_reg_params
from . import train_common as tc


def initialize_reg_params(model):
    """train_SI.py:286-298."""
    reg_params = {}
    for name, param in model.named_parameters():
        reg_params[param] = {"omega": torch.zeros_like(param.data), "w": torch.zeros_like(param.data),
                             "init_val": param.data.clone(), "name": name}
    return reg_params


def update_reg_params(model, slak=1e-3):
    """train_SI.py:301-364 (== the redefinition at :367-430): for known params
    omega += max(w / ((theta - init)^2 + slak), 0); w <- 0; init <- theta (clhip_si_consolidate);
    unknown params (fresh head) get zeros."""
    reg_params = model.reg_params
    for param in list(model.parameters()):
        if param in reg_params:
            rp = reg_params[param]
            dev = param.device
            omega = rp["omega"].to(dev).contiguous()
            w = rp["w"].to(dev).contiguous()
            init_val = rp["init_val"].to(dev).contiguous().clone()
            ops.si_consolidate(omega, w, param.data.contiguous(), init_val, slak)
            rp["omega"], rp["w"], rp["init_val"] = omega, w, init_val
        else:
            reg_params[param] = {"omega": torch.zeros_like(param.data), "w": torch.zeros_like(param.data),
                                 "init_val": param.data.clone()}
    return reg_params


def fine_tune_elastic(dataset_path, model_path, exp_dir, batch_size=200, num_epochs=100, lr=0.0004, reg_lambda=100,
                      init_freeze=0, weight_decay=0, saving_freq=5, device="cuda"):
    """main_SI.py:26-94."""
    dsets = load_task_datasets(dataset_path)
    dset_loaders = tc.make_loaders(dsets, batch_size, device)
    dset_sizes = {x: len(dsets[x]) for x in ["train", "val"]}
    dset_classes = dsets["train"].classes
    resume = os.path.join(exp_dir, "epoch.pth.tar")
    resumed = os.path.isfile(resume)
    if resumed:
        model_ft = torch.load(resume, weights_only=False)["model"]
    else:
        if not os.path.isfile(model_path):
            raise FileNotFoundError("model_path %r (the reference falls back to a pretrained AlexNet download here, "
                                    "main_SI.py:44-48; no network on this path)" % model_path)
        model_ft = tc.load_model(model_path)
        if not init_freeze:
            tc.replace_head(model_ft, len(dset_classes))
        os.makedirs(exp_dir, exist_ok=True)
    model_ft = model_ft.to(device)
    t0 = time.time()
    if not resumed:
        if not hasattr(model_ft, "reg_params"):
            reg_params = initialize_reg_params(model_ft)
        else:
            parameters = list(model_ft.parameters())
            model_ft.reg_params.pop(parameters[-1], None)      # main_SI.py:73-77 (no-op after head swap)
            model_ft.reg_params.pop(parameters[-2], None)
            model_ft.reg_params.pop("lambda", None)
            reg_params = update_reg_params(model_ft)
        reg_params["lambda"] = reg_lambda
        model_ft.reg_params = reg_params
    tc.save_preprocessing_time(exp_dir, time.time() - t0)
    engine = tc.engine_for(model_ft, dset_loaders, batch_size, device)
    arena_reg_params(engine.arena, model_ft.reg_params, names=("omega", "init_val", "w"))
    optimizer_ft = Elastic_SGD(model_ft.parameters(), lr, momentum=0.9, weight_decay=weight_decay)
    # SI: range(start, num_epochs + 1) and stop at count >= 10 (train_SI.py:182,132)
    return tc.train_model(model_ft, engine, optimizer_ft, lr, dset_loaders, dset_sizes, num_epochs, exp_dir, resume,
                          saving_freq=saving_freq, early_stop="ge", extra_epoch=True)
