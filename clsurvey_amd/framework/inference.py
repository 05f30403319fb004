"""Test-time evaluation — mirror of src/framework/inference.py:8-87 and utils.get_prev_heads
(utilities/utils.py:235-262): swap the task head in, forward the test split, top-1 accuracy x100."""
import copy
from types import SimpleNamespace

import torch

from ..data import DeviceLoader, load_task_datasets
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
    """Top-1 accuracy (percent) of `model` on one split of a task file, through `method.get_output(images, holder)` —
    the contract of framework/inference.py:8-87: `holder` carries the model, the separately saved heads (`target_head`,
    in which case the head index addresses that list) or the wrapper's own head index, and the classifier's last slot.
    Images stay on the device; hits are counted there and read once."""
    heads = None if target_head is None else (target_head if isinstance(target_head, list) else [target_head])
    if heads is not None:                                    # inference.py:15-18
        assert target_task_head_idx == 0, "Only EBLL, LWF have heads in model itself, here head idx indicates target_headlist idx"
    if hasattr(model, "classifier"):
        final_layer_idx = str(len(model.classifier._modules) - 1)
    model.eval()
    model = model.to(device)
    dsets = load_task_datasets(dataset_path)
    split = subset if "test" in dsets else "val"            # a task file without a test split is scored on val (inference.py:27-33)
    holder = SimpleNamespace(task_imgfolders=dsets, batch_size=batch_size, model=model, heads=heads,
                             current_head_idx=target_task_head_idx, final_layer_idx=final_layer_idx, task_idx=task_idx)
    hits = torch.zeros((), dtype=torch.int64, device=device)
    seen = 0
    for images, labels in DeviceLoader(dsets[split], batch_size, True, device):
        hits += (method.get_output(images, holder).argmax(1) == labels).sum()
        seen += labels.shape[0]
    accuracy = 100.0 * float(hits.item()) / seen
    print("Overall Accuracy: " + str(accuracy))
    return accuracy
