#!/usr/bin/env python
"""As coresident_pair.py (same launches on one stream / on separate streams without events), for the remaining candidates of a
merged grid in small_VGG9's backward pass at N = 200: the weight gradients of layers 2 and 1 (the second needs only the backward-data
of layer 2), and the first layer's weight gradient beside the slab reduction of the other layers.
usage: python tools/experiments/coresident_more.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clsurvey_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
N = 200
SPIN = 40_000_000


def timed(branches):
    streams = [torch.cuda.Stream(device=dev) for _ in branches]
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    hold = torch.cuda.Event()
    with torch.cuda.stream(streams[0]):
        torch.cuda._sleep(SPIN)
        hold.record()
        e0.record()
    ends = []
    for s, fns in zip(streams, branches):
        with torch.cuda.stream(s):
            s.wait_event(hold)
            for _ in range(reps):
                for f in fns:
                    f()
            ev = torch.cuda.Event()
            ev.record()
            ends.append(ev)
    with torch.cuda.stream(streams[0]):
        for ev in ends:
            streams[0].wait_event(ev)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def report(name, a, b):
    for _ in range(2):
        ta = timed([[a]]); tb = timed([[b]]); s = timed([[a, b]]); c = timed([[a], [b]])
    print("%-58s A %6.1f  B %6.1f  one stream %6.1f  two streams %6.1f  (%+.1f %%)" % (name, ta, tb, s, c, (c / s - 1) * 100))


x1 = torch.randn(N, 3, 64, 64, device=dev)
w1 = torch.randn(64, 3, 3, 3, device=dev) * 0.1
b1 = torch.zeros(64, device=dev)
y1, idx1 = ops.conv3x3_relu_pool_fwd(x1, w1, b1)                      # [N, 64, 32, 32]
dy1 = torch.randn_like(y1)
w2 = torch.randn(64, 64, 3, 3, device=dev) * 0.05
y2, idx2 = ops.conv3x3_wino_fwd(y1, w2, b1, True, pool=True)          # [N, 64, 16, 16]
dy2 = torch.randn_like(y2)

W1 = lambda: ops.conv3x3_bwd_weight_unpool(x1, dy1, idx1)             # noqa: E731  first layer, fused un-pooling (c3 kernel + its reduction)
W2 = lambda: ops.conv3x3_wino_bwd_weight(y1, dy2, idx2)               # noqa: E731  layer 2 (pixel-split Winograd kernel + its reduction)
D2 = lambda: ops.conv3x3_bs_bwd_data(dy2, w2, None, idx2)             # noqa: E731
slabs = torch.randn(128 * (9 * 64 * 64 + 64) * 2, device=dev).view(torch.uint8)
R = lambda: (ops.conv3x3_bwd_weight_reduce(slabs, 256, 64, 64), ops.conv3x3_bwd_weight_reduce(slabs, 256, 64, 64))   # noqa: E731  ~ 75 MB of slabs
for f in (W1, W2, D2, R):
    f()
torch.cuda.synchronize()
print("us per round, N = %d, %d rounds" % (N, reps))
report("A = wgrad layer 2 (64->64 @32x32 + pool), B = wgrad layer 1", W2, W1)
report("A = bwd-data layer 2 (bf16-split), B = wgrad layer 1", D2, W1)
report("A = wgrad layer 1, B = slab reduction (2 x 37 MB)", W1, R)
report("A = wgrad layer 2, B = slab reduction (2 x 37 MB)", W2, R)
