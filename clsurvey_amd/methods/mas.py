"""MAS on the HIP path — mirror of src/methods/MAS/{main_MAS,train_MAS}.py (only the reachable
configuration norm='L2', b1=False: methods/method.py:748)."""
import os
import time

import torch.nn as nn

from ..data import DeviceLoader, load_task_datasets
from ..net import NetEngine
from ..optim import Objective_After_SGD, Weight_Regularized_SGD, arena_reg_params
from . import train_common as tc
from . import ewc as _ewc

initialize_reg_params = _ewc.initialize_reg_params          # train_MAS.py:694-705 (same body)
initialize_store_reg_params = _ewc.store_prev_reg_params    # train_MAS.py:708-733
accumelate_reg_params = _ewc.accumelate_reg_params          # train_MAS.py:769-795


def compute_importance_l2(model, optimizer, lr_scheduler, dset_loaders, use_gpu=True, engine=None):
    """train_MAS.py:508-567: per batch loss = sum(outputs^2) (MSELoss(size_average=False) vs zeros),
    backward, then Objective_After_SGD.step(reg_params, index, labels.size(0)) — the batch size
    used in the running mean is the CURRENT batch's (short-last-batch quirk kept)."""
    reg_params = model.reg_params
    model.eval()                                # train_MAS.py:518
    first = dset_loaders[0]
    if engine is None:
        engine = NetEngine(model, first.batch_size, tuple(first.x.shape[1:]), first.device)
    A = engine.arena
    optimizer._arena = A
    arena_reg_params(A, reg_params, names=("omega",))
    index = 0
    for dset_loader in dset_loaders:
        for inputs, labels in dset_loader:
            engine.loss_step(inputs, None, "mse_sum_zero", backward=True)
            optimizer.step(reg_params, index, labels.size(0))
            index += 1
    for p in A.params:
        if p in reg_params:
            reg_params[p]["omega"] = A.view("omega", p).clone()
    reg_params.pop("__arena__", None)
    return model


def accumulate_objective_based_weights(data_dir, reg_sets, model_ft, batch_size, norm="L2", test_set="train",
                                       device="cuda"):
    """main_MAS.py:109-153."""
    if norm != "L2":
        raise NotImplementedError("only norm='L2' is reachable from the framework (method.py:748)")
    dset_loaders = []
    for data_path in reg_sets:
        dset = load_task_datasets(data_path)
        dset_loaders.append(DeviceLoader(dset[test_set], batch_size, False, device))
    if not hasattr(model_ft, "reg_params"):
        model_ft.reg_params = initialize_reg_params(model_ft)
    model_ft.reg_params = initialize_store_reg_params(model_ft)
    optimizer_ft = Objective_After_SGD(model_ft.parameters(), lr=0.0001, momentum=0.9)
    model_ft = compute_importance_l2(model_ft, optimizer_ft, None, dset_loaders)
    model_ft.reg_params = accumelate_reg_params(model_ft)
    return model_ft


def fine_tune_objective_based_acuumelation(dataset_path, previous_task_model_path, init_model_path, exp_dir, data_dir,
                                           reg_sets, reg_lambda=1, norm="L2", num_epochs=100, lr=0.0008,
                                           batch_size=200, weight_decay=0, b1=True, L1_decay=False, head_shared=False,
                                           saving_freq=5, device="cuda"):
    """main_MAS.py:34-106."""
    if L1_decay:
        raise NotImplementedError("L1_decay is never set on the framework path")
    dsets = load_task_datasets(dataset_path)
    dset_loaders = tc.make_loaders(dsets, batch_size, device)
    dset_sizes = {x: len(dsets[x]) for x in ["train", "val"]}
    dset_classes = dsets["train"].classes
    t0 = time.time()
    model_ft = tc.load_model(previous_task_model_path)
    if isinstance(model_ft, dict):
        model_ft = model_ft["model"]
    model_ft = model_ft.to(device)
    update_batch_size = 1 if b1 else batch_size
    model_ft = accumulate_objective_based_weights(data_dir, reg_sets, model_ft, update_batch_size, norm, "train", device)
    model_ft.reg_params["lambda"] = reg_lambda
    tc.save_preprocessing_time(exp_dir, time.time() - t0)
    if not head_shared:
        last = str(len(model_ft.classifier._modules) - 1)
        if init_model_path is not None:
            init_model = tc.load_model(init_model_path)
            model_ft.classifier._modules[last] = init_model.classifier._modules[last]
        else:
            num_ftrs = model_ft.classifier._modules[last].in_features
            model_ft.classifier._modules[last] = nn.Linear(num_ftrs, len(dset_classes))
    model_ft = model_ft.to(device)
    engine = tc.engine_for(model_ft, dset_loaders, batch_size, device)
    arena_reg_params(engine.arena, model_ft.reg_params)
    optimizer_ft = Weight_Regularized_SGD(model_ft.parameters(), lr, momentum=0.9, weight_decay=weight_decay)
    os.makedirs(exp_dir, exist_ok=True)
    resume = os.path.join(exp_dir, "epoch.pth.tar")
    return tc.train_model(model_ft, engine, optimizer_ft, lr, dset_loaders, dset_sizes, num_epochs, exp_dir, resume,
                          saving_freq=saving_freq)
