"""Test-time evaluation — mirror of src/framework/inference.py:8-87 and utils.get_prev_heads
(utilities/utils.py:235-262): swap the task head in, forward the test split, top-1 accuracy x100."""
import copy

import torch
from ..data import load_task_datasets

from ..data import DeviceLoader
from ..methods import train_common as tc


def get_prev_heads(prev_head_model_paths, head_layer_idx, device="cuda"):
    if not isinstance(prev_head_model_paths, list):
        prev_head_model_paths = [prev_head_model_paths]
    heads = []
    for path in prev_head_model_paths:
        m = tc.load_model(path)
        if isinstance(m, dict):
            m = m["model"]
        head = m.classifier._modules[head_layer_idx]
        assert isinstance(head, torch.nn.Linear), type(head)
        heads.append(copy.deepcopy(head.to(device)))
    return heads


def test_model(method, model, dataset_path, target_task_head_idx, target_head=None, batch_size=200, subset="test",
               per_class_stats=False, final_layer_idx=None, task_idx=None, device="cuda"):
    if target_head is not None and not isinstance(target_head, list):
        target_head = [target_head]
    if hasattr(model, "classifier"):
        final_layer_idx = str(len(model.classifier._modules) - 1)
    model.eval()
    model = model.to(device)
    dsets = load_task_datasets(dataset_path)
    if subset not in dsets:
        subset = "val"
    loader = DeviceLoader(dsets[subset], batch_size, True, device)
    holder = type("Holder", (object,), {})()
    holder.task_imgfolders = dsets
    holder.batch_size = batch_size
    holder.model = model
    holder.heads = target_head
    holder.current_head_idx = target_task_head_idx
    holder.final_layer_idx = final_layer_idx
    holder.task_idx = task_idx
    correct = torch.zeros((), dtype=torch.int64, device=device)
    total = 0
    for images, labels in loader:
        outputs = method.get_output(images, holder)
        correct += (outputs.argmax(1) == labels).sum()
        total += labels.shape[0]
    accuracy = float(correct.item()) * 100.0 / total
    print("Overall Accuracy: " + str(accuracy))
    return accuracy
