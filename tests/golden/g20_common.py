"""Deterministic inputs shared by the G20 fixture generator (reference run, dev container) and the GPU tests of
config 5 at its real widths (wide_VGG9_cl_512_512): numpy-driven parameters by (order, shape), batches, PackNet
owner masks, and the sampling that keeps the fixture small (4.7 M - 56 M parameters per model are NOT stored: both
sides regenerate them from the seed; the fixture holds the reference's OUTPUTS at sampled positions + checksums)."""
import hashlib

import numpy as np

WIDE = [64, "M", 128, "M", 256, 256, "M", 512, 512, "M"]
FC = (512, 512)
NCLS = 20
FULL_BELOW = 1 << 16
NSAMPLE = 8192


def fill_params(named_shapes, seed):
    """[(name, shape)] in named_parameters() order -> list of float32 arrays: kaiming-normal weights, small non-zero
    biases (so that the bias-freezing rules are exercised), HAT embeddings uniform in [-1.5, 2]."""
    gen = np.random.RandomState(seed)
    out = []
    for name, shape in named_shapes:
        shape = tuple(int(s) for s in shape)
        if "embs" in name:
            out.append(gen.uniform(-1.5, 2.0, size=shape).astype(np.float32))
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            out.append((gen.standard_normal(shape) * (2.0 / fan_in) ** 0.5).astype(np.float32))
        else:
            out.append((0.05 * gen.standard_normal(shape)).astype(np.float32))
    return out


def batch(seed, n, hw, ncls=NCLS):
    gen = np.random.RandomState(seed)
    return gen.standard_normal((n, 3, hw, hw)).astype(np.float32), gen.randint(0, ncls, size=(n,)).astype(np.int64)


def owner_mask(seed, shape):
    """PackNet owner index per weight: 0 = pruned / free, 1 = the earlier task's, 2 = the current task's."""
    return np.random.RandomState(seed).randint(0, 3, size=tuple(shape)).astype(np.uint8)


def positions(numel, seed):
    if numel <= FULL_BELOW:
        return np.arange(numel)
    return np.sort(np.random.RandomState(seed).randint(0, numel, size=NSAMPLE))


def digest(a, seed):
    """what the fixture keeps of one tensor: sampled values, float64 sum / sum of magnitudes, element count."""
    flat = np.asarray(a).reshape(-1)
    return dict(v=flat[positions(flat.size, seed)].copy(), s=np.array([flat.astype(np.float64).sum(), np.abs(flat.astype(np.float64)).sum(), flat.size]))


def zero_pattern(a):
    """sha256 over the packed (value == 0) bitmap: PackNet's pruned positions, bit-exact."""
    return hashlib.sha256(np.packbits(np.asarray(a).reshape(-1) == 0).tobytes()).hexdigest()
