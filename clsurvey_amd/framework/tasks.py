"""Task-sequence dataset objects with the interface of src/data/dataset.py CustomDataset
(name, argname, test_results_dir, train_exp_results_dir, task_count, classes_per_task, input_size,
get_task_dataset_path(task_name, rnd_transform), get_taskname(i)).

There is no Tiny-ImageNet in the container (no network), so `SyntheticTinyImagenet` writes tensor
tasks of the same shape (10 tasks x 20 classes, 8000/2000/1000 images of 3x64x64,
data/tinyimgnet_dataprep.py:69-151) — class-conditional Gaussian prototypes + noise so accuracies
and forgetting are non-trivial — as pickled {'train','val','test'} dicts, the same wire format the
reference's framework passes between its layers."""
import json
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

    def spec(self, task_name):
        """Everything the bytes of a task file depend on."""
        return {"sizes": [int(v) for v in self.sizes], "classes": int(self.n_classes), "hw": int(self.hw),
                "seed": int(self.seed) * 1000 + int(task_name), "noise": float(self.noise), "kind": str(self.kind),
                "blobs": None if self.blobs is None else {k: float(v) for k, v in sorted(dict(self.blobs).items())}}

    def get_task_dataset_path(self, task_name=None, rnd_transform=False):
        """Task files are cached under root/name/; a sidecar task_N.spec.json records what they were generated from.  A cached
        file of ANOTHER spec (other kind / blobs / noise / sizes / seed under the same results root) is an error, not a hit:
        the results tree beside it holds success tokens and models of that other data."""
        path = os.path.join(self.root, self.name, "task_%s.pth.tar" % task_name)
        side = os.path.join(self.root, self.name, "task_%s.spec.json" % task_name)
        want = self.spec(task_name)
        if os.path.exists(path) and os.path.exists(side):
            with open(side) as f:
                have = json.load(f)
            if have != want:
                raise RuntimeError("%s was generated from %s, this run asks for %s: use a fresh --results_root (its results tree "
                                   "belongs to the other data)" % (path, have, want))
            return path
        os.makedirs(os.path.dirname(path), exist_ok=True)
        if os.path.exists(path) and not os.path.exists(side):
            # data without its spec (a cache older than the sidecars, or a writer that died between the two files): regenerating is
            # only safe while no results tree (the driver keeps <results_root>/train and /test beside <results_root>/data) was
            # built from the old bytes
            if any(os.path.isdir(os.path.join(os.path.dirname(self.root), d)) for d in ("train", "test")):
                raise RuntimeError("%s has no %s beside it but a results tree exists under %s: cannot tell what data those results "
                                   "were made from — use a fresh --results_root" % (path, os.path.basename(side), os.path.dirname(self.root)))
        d = synthetic_task(self.sizes[0], self.sizes[1], self.sizes[2], self.n_classes, self.hw,
                           seed=want["seed"], noise=self.noise, kind=self.kind, blobs=self.blobs)
        tmp = "%s.tmp.%d" % (path, os.getpid())    # (torch.save is not atomic: a killed writer must not leave a truncated task file)
        torch.save(d, tmp)
        os.replace(tmp, path)
        tmp = "%s.tmp.%d" % (side, os.getpid())
        with open(tmp, "w") as f:
            json.dump(want, f)
        os.replace(tmp, side)                      # written last: a data file without a sidecar is never a hit
        return path
