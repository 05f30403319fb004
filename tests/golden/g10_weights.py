"""Deterministic start weights shared by the G10 fixture generator (reference run, dev container) and
the GPU end-to-end test (build's driver): numpy-driven, kaiming for conv AND linear so that the tiny
tasks are learnable in 8 epochs."""
import numpy as np

SMALL = [64, "M", 64, "M", 64, 64, "M", 128, 128, "M"]


def det_weights(seed=5, cfg=None, fc=(128, 128), ncls=4, hw=32):
    """deterministic, quickly-trainable start weights (kaiming for conv AND linear), numpy-driven."""
    gen = np.random.RandomState(seed)
    cfg = SMALL if cfg is None else cfg
    ws, c = [], 3
    for v in cfg:
        if v == "M":
            hw //= 2
            continue
        ws.append((gen.standard_normal((v, c, 3, 3)) * (2.0 / (v * 9)) ** 0.5).astype(np.float32))
        ws.append(np.zeros(v, dtype=np.float32))
        c = v
    d = c * hw * hw
    for o in (fc[0], fc[1], ncls):
        ws.append((gen.standard_normal((o, d)) * (2.0 / d) ** 0.5).astype(np.float32))
        ws.append(np.zeros(o, dtype=np.float32))
        d = o
    return ws


