#!/usr/bin/env python
"""Would ONE grid for (backward-data + weight gradient) of a small layer, or for the four small weight gradients, pay?
Upper bound without building the merged kernel: the same launches issued on separate HIP streams with NO event between them
(a spin kernel holds both streams while the host enqueues, so the device sees back-to-back work on every stream), against
the same launches on one stream.  What the hardware does with two resident kernels is exactly what it would do with the two
halves of a merged grid (block-level co-residency on the CUs); the cross-stream events of CLHIP_WGRAD_OVERLAP are not there.
usage: python tools/experiments/coresident_pair.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clsurvey_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
N = 200
SPIN = 40_000_000          # cycles the blocker holds the streams (about 20 ms): longer than the host needs to enqueue


def layer(C, K, HW, pooled, bs=False):
    x = torch.randn(N, C, HW, HW, device=dev)
    w = torch.randn(K, C, 3, 3, device=dev) * 0.05
    b = torch.zeros(K, device=dev)
    if pooled:
        yp, idx = ops.conv3x3_wino_fwd(x, w, b, True, pool=True)
        dyp = torch.randn_like(yp)
        if bs:
            return (lambda: ops.conv3x3_bs_bwd_data(dyp, w, None, idx)), (lambda: ops.conv3x3_wino_bwd_weight(x, dyp, idx))
        return (lambda: ops.conv3x3_wino_bwd_data(dyp, w, None, idx)), (lambda: ops.conv3x3_wino_bwd_weight(x, dyp, idx))
    dy = torch.randn(N, K, HW, HW, device=dev)
    return (lambda: ops.conv3x3_wino_bwd_data(dy, w, x)), (lambda: ops.conv3x3_wino_bwd_weight(x, dy))


def timed(branches):
    """branches: list of lists of callables; branch i runs `reps` rounds of its callables on its own stream.  Returns
    microseconds per round (all branches together)."""
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


shapes = [("64->64 @16x16", 64, 64, 16, False), ("64->64 @16x16 + pool", 64, 64, 16, True),
          ("64->128 @8x8", 64, 128, 8, False), ("128->128 @8x8 + pool", 128, 128, 8, True)]
pairs = [layer(C, K, HW, p) for _, C, K, HW, p in shapes]
for d, wg in pairs:                       # warm up (allocator pools of every stream are filled inside timed(), first call)
    d(); wg()
torch.cuda.synchronize()
print("us per round, N = %d, %d rounds; one stream / separate streams without events" % (N, reps))
tot_serial = tot_conc = 0.0
for (name, *_), (d, wg) in zip(shapes, pairs):
    for _ in range(2):                    # second pass is the one reported (stream-private allocator pools warm)
        a = timed([[d]]); b = timed([[wg]]); s = timed([[d, wg]]); c = timed([[d], [wg]])
    tot_serial += s; tot_conc += c
    print("%-22s bwd-data %6.1f  wgrad %6.1f  both, one stream %6.1f  both, two streams %6.1f  (%+.1f %%)"
          % (name, a, b, s, c, (c / s - 1) * 100))
d2, w2 = layer(64, 64, 32, True, bs=True)
d2(); w2()
for _ in range(2):
    a = timed([[d2]]); b = timed([[w2]]); s = timed([[d2, w2]]); c = timed([[d2], [w2]])
print("%-22s bwd-data %6.1f  wgrad %6.1f  both, one stream %6.1f  both, two streams %6.1f  (%+.1f %%)   [bf16-split bwd-data + Winograd wgrad]"
      % ("64->64 @32x32 + pool", a, b, s, c, (c / s - 1) * 100))
print("four layers: one stream %.1f, pairs co-resident %.1f (%+.1f %%)" % (tot_serial, tot_conc, (tot_conc / tot_serial - 1) * 100))
wgs = [wg for _, wg in pairs]
ds = [d for d, _ in pairs]
for _ in range(2):
    s4 = timed([wgs]); c4 = timed([[f] for f in wgs])
    s8 = timed([ds + wgs]); c8 = timed([[f] for f in ds + wgs])
print("the four weight gradients: one stream %.1f, four streams %.1f (%+.1f %%)" % (s4, c4, (c4 / s4 - 1) * 100))
print("all eight launches: one stream %.1f, eight streams %.1f (%+.1f %%)" % (s8, c8, (c8 / s8 - 1) * 100))
