"""Device-resident task data.

The reference trains on pickled ImageFolder datasets that re-decode 8000 JPEGs per epoch through
8 DataLoader workers (EWC/main_EWC.py:28-31, data/imgfolder.py:86-128).  The framework uses the
raw, un-augmented sets (framework/main.py:197-202), so a whole Tiny-ImageNet task is a static
393 MB tensor: here it lives in HBM once and batches are gathered on device.

`DeviceLoader` reproduces the batch composition of torch's DataLoader(shuffle=True) bit for bit
for the same global torch RNG state (it consumes the global generator exactly like
_BaseDataLoaderIter.__init__ + RandomSampler.__iter__ do), so accuracy traces can be compared with
the reference run under the same seed (utilities/utils.py:52-58).
"""
import torch
from torch.utils.data import Dataset


class TensorTaskDataset(Dataset):
    """One split of one task. `classes` mirrors ImageFolder_Subset.classes (data/imgfolder.py)."""

    def __init__(self, x, y, classes):
        assert x.shape[0] == y.shape[0]
        self.x = x.contiguous().float()
        self.y = y.contiguous().long()
        self.classes = list(classes)

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return self.x[i], self.y[i]


def _extract(dataset):
    """(x, y) tensors of any map-style dataset (fast path for TensorTaskDataset)."""
    if isinstance(dataset, TensorTaskDataset):
        return dataset.x, dataset.y
    xs, ys = [], []
    for i in range(len(dataset)):
        item = dataset[i]
        xs.append(torch.as_tensor(item[0]))
        ys.append(int(item[1]))
    return torch.stack(xs).float(), torch.tensor(ys, dtype=torch.int64)


class DeviceLoader:
    """Iterates (x, y) batches of a dataset held in HBM. Same length / order semantics as
    torch.utils.data.DataLoader(dataset, batch_size, shuffle, drop_last=False)."""

    def __init__(self, dataset, batch_size, shuffle, device="cuda"):
        self.dataset = dataset
        self.batch_size = int(batch_size)
        self.shuffle = bool(shuffle)
        self.device = torch.device(device)
        x, y = _extract(dataset)
        self.x = x.to(self.device)
        self.y = y.to(self.device)
        self.n = self.x.shape[0]

    def __len__(self):
        return (self.n + self.batch_size - 1) // self.batch_size

    def order(self):
        # _BaseDataLoaderIter.__init__ draws the worker base seed first ...
        torch.empty((), dtype=torch.int64).random_()
        if not self.shuffle:
            return None
        # ... then RandomSampler.__iter__ seeds a private generator from the global one
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        g = torch.Generator()
        g.manual_seed(seed)
        return torch.randperm(self.n, generator=g)

    def __iter__(self):
        perm = self.order()
        if perm is not None:
            perm = perm.to(self.device)
        for s in range(0, self.n, self.batch_size):
            if perm is None:
                yield self.x[s:s + self.batch_size], self.y[s:s + self.batch_size]
            else:
                idx = perm[s:s + self.batch_size]
                yield self.x.index_select(0, idx), self.y.index_select(0, idx)


def synthetic_task(n_train, n_val, n_test, n_classes, hw=64, seed=7, noise=1.0, device="cpu"):
    """Learnable synthetic task (class-conditional Gaussian prototypes + noise), SURVEY §8d."""
    g = torch.Generator()
    g.manual_seed(seed)
    protos = torch.randn((n_classes, 3, hw, hw), generator=g) * 0.5
    out = {}
    for name, n in (("train", n_train), ("val", n_val), ("test", n_test)):
        y = torch.randint(0, n_classes, (n,), generator=g)
        x = protos[y] + noise * torch.randn((n, 3, hw, hw), generator=g)
        out[name] = TensorTaskDataset(x.to(device), y.to(device), [str(c) for c in range(n_classes)])
    return out
