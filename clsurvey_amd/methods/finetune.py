"""Plain fine-tuning on the HIP path — mirror of src/methods/Finetune/{main_SGD,train_SGD}.py.
Phase 1 of every framework method delegates here (methods/method.py:671,706,735)."""
import os

import torch

from ..optim import SGD
from . import train_common as tc


def fine_tune_SGD(dset_dataloader, cumsum_dset_sizes, dset_classes, model_path, exp_dir, num_epochs=100, lr=0.0004,
                  freeze_mode=0, weight_decay=0, enable_resume=True, replace_last_classifier_layer=True,
                  save_models_mode=True, freq=5, device="cuda", batch_size=None):
    """main_SGD.py:13-82. dset_dataloader: {'train','val'} of DeviceLoader (Finetune.grid_datafetch
    builds them, method.py:1030-1060)."""
    resume = os.path.join(exp_dir, "epoch.pth.tar") if enable_resume else ""
    if resume and os.path.isfile(resume):
        model_ft = torch.load(resume, weights_only=False)["model"]
    else:
        if not os.path.exists(exp_dir) and save_models_mode:
            os.makedirs(exp_dir)
        if not os.path.isfile(model_path):
            raise Exception("Model path non-existing: {}".format(model_path))
        model_ft = tc.load_model(model_path)
    # main_SGD.py:50-53: an LwF wrapper is unwrapped to its VGG with the stacked heads cut down to the first one
    from .lwf import AlexNet_LwF
    if isinstance(model_ft, AlexNet_LwF):
        model_ft.model.classifier = torch.nn.Sequential(
            *list(model_ft.model.classifier.children())[:model_ft.last_layer_name + 1])
        model_ft = model_ft.model
    from .ebll import AlexNet_EBLL
    engine_params = None
    if isinstance(model_ft, AlexNet_EBLL):                        # main_SGD.py:54-56: stays a wrapper, in finetune mode
        model_ft.classifier = torch.nn.Sequential(*list(model_ft.classifier.children())[:model_ft.last_layer_name + 1])
        model_ft.set_finetune_mode(True)
    if freeze_mode or replace_last_classifier_layer:            # main_SGD.py:59
        labels_per_task = [len(task_labels) for task_labels in dset_classes["train"]]
        tc.replace_head(model_ft, sum(labels_per_task))          # utils.py:68-72
    model_ft = model_ft.to(device)
    any_loader = dset_dataloader["train"]
    if isinstance(model_ft, AlexNet_EBLL):
        # the frozen encoders get no gradient in the reference (p.grad is None: SGD skips them, weight decay included):
        # keep them out of the arena the fused SGD kernel sweeps
        from ..net import NetEngine
        engine_params = list(model_ft.features.parameters()) + list(model_ft.classifier.parameters())
        engine = NetEngine(model_ft, batch_size or any_loader.batch_size, tuple(any_loader.x.shape[1:]), device,
                           params=engine_params)
    else:
        engine = tc.engine_for(model_ft, dset_dataloader, batch_size or any_loader.batch_size, device)
    if freeze_mode:
        # main_SGD.py:69-72: warm-up of the fresh head — the optimizer sees the last classifier module only, without
        # weight decay (the reference names it classifier['6'], the head slot of its 7-module classifiers); the plan
        # executor still produces every gradient, the rest of the arena is simply never stepped
        last = str(len(model_ft.classifier._modules) - 1)
        optimizer_ft = SGD(model_ft.classifier._modules[last].parameters(), lr, momentum=0.9)
    else:
        optimizer_ft = SGD(engine_params if engine_params is not None else model_ft.parameters(), lr, momentum=0.9,
                           weight_decay=weight_decay)
    return tc.train_model(model_ft, engine, optimizer_ft, lr, dset_dataloader, cumsum_dset_sizes, num_epochs, exp_dir,
                          resume, saving_freq=freq, step_fn=optimizer_ft.step, save_models_mode=save_models_mode,
                          abort_on_bad_loss=False)
