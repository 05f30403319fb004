"""Task-sequence dataset objects with the interface of src/data/dataset.py CustomDataset
(name, argname, test_results_dir, train_exp_results_dir, task_count, classes_per_task, input_size,
get_task_dataset_path(task_name, rnd_transform), get_taskname(i)).

There is no Tiny-ImageNet in the container (no network), so `SyntheticTinyImagenet` writes tensor
tasks of the same shape (10 tasks x 20 classes, 8000/2000/1000 images of 3x64x64,
data/tinyimgnet_dataprep.py:69-151) — class-conditional Gaussian prototypes + noise so accuracies
and forgetting are non-trivial — as pickled {'train','val','test'} dicts, the same wire format the
reference's framework passes between its layers."""
import os
from collections import OrderedDict

import torch

from ..data import synthetic_task


class SyntheticTaskSequence(object):
    def __init__(self, root, task_count=10, classes_per_task=20, sizes=(8000, 2000, 1000), hw=64, seed=7, noise=1.0,
                 name="synthetic_tiny_imagenet", kind="protos", blobs=None):
        self.name = name
        self.argname = name
        self.test_results_dir = name
        self.train_exp_results_dir = name
        self.task_count = task_count
        self.input_size = (hw, hw)
        self.classes_per_task = OrderedDict((self.get_taskname(i), [str(c) for c in range(classes_per_task)])
                                            for i in range(1, task_count + 1))
        self.root = root
        self.sizes = sizes
        self.hw = hw
        self.seed = seed
        self.noise = noise
        self.kind = kind
        self.blobs = blobs
        self.n_classes = classes_per_task

    def get_taskname(self, task_index):
        return str(task_index)

    def get_task_dataset_path(self, task_name=None, rnd_transform=False):
        path = os.path.join(self.root, self.name, "task_%s.pth.tar" % task_name)
        if not os.path.exists(path):
            os.makedirs(os.path.dirname(path), exist_ok=True)
            d = synthetic_task(self.sizes[0], self.sizes[1], self.sizes[2], self.n_classes, self.hw,
                               seed=self.seed * 1000 + int(task_name), noise=self.noise, kind=self.kind, blobs=self.blobs)
            torch.save(d, path)
        return path
