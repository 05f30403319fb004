#!/usr/bin/env python
"""The forward launches of small_VGG9's deep layers have no independent sibling inside a pass — unless the batch is cut in halves:
(layer L on half B) is independent of (layer L + 1 on half A).  Bound without building merged kernels: the chain of the four deep
forward launches at N = 200 on one stream, against two half-batch chains (N = 100 each) on two streams without events, the second
held back by a spin of `lag` microseconds so that the two chains are one layer apart.  The same for the merged backward operators.
usage: python tools/experiments/coresident_halves.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clsurvey_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
SPIN = 40_000_000
CYC_PER_US = 100            # torch.cuda._sleep counts a 100 MHz clock on this part (calibrated below)


def chain(N):
    """closures of the four deep forward launches and of the four merged backward operators on a batch of N"""
    L = [(64, 64, 16, False), (64, 64, 16, True), (64, 128, 8, False), (128, 128, 8, True)]
    fw, bw = [], []
    for C, K, HW, pool in L:
        x = torch.randn(N, C, HW, HW, device=dev)
        w = torch.randn(K, C, 3, 3, device=dev) * 0.05
        b = torch.zeros(K, device=dev)
        if pool:
            yp, idx = ops.conv3x3_wino_fwd(x, w, b, True, pool=True)
            dy = torch.randn_like(yp)
            fw.append(lambda x=x, w=w, b=b: ops.conv3x3_wino_fwd(x, w, b, True, pool=True))
            bw.append(lambda x=x, w=w, dy=dy, idx=idx: ops.conv3x3_wino_bwd(x, dy, w, None, idx))
        else:
            dy = torch.randn(N, K, HW, HW, device=dev)
            fw.append(lambda x=x, w=w, b=b: ops.conv3x3_wino_fwd(x, w, b, True))
            bw.append(lambda x=x, w=w, dy=dy: ops.conv3x3_wino_bwd(x, dy, w, x, None))
    return fw, bw[::-1]


def timed(branches, lags):
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
    for s, fns, lag in zip(streams, branches, lags):
        with torch.cuda.stream(s):
            s.wait_event(hold)
            if lag:
                torch.cuda._sleep(int(lag * CYC_PER_US))
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


# calibrate the spin clock: 1e6 counts in microseconds
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000); torch.cuda.synchronize()
a.record(); torch.cuda._sleep(1_000_000); b.record(); torch.cuda.synchronize()
CYC_PER_US = 1_000_000 / (a.elapsed_time(b) * 1e3)
print("spin clock: %.0f counts per us" % CYC_PER_US)

f200, b200 = chain(200)
fa, ba = chain(100)
fb, bb = chain(100)
for f in f200 + b200 + fa + ba + fb + bb:
    f()
torch.cuda.synchronize()
for name, full, ha, hb in (("forward of the four deep layers", f200, fa, fb), ("merged backward operators of the four deep layers", b200, ba, bb)):
    for _ in range(2):
        t_full = timed([full], [0])
        t_half = timed([ha], [0])
        res = {lag: timed([ha, hb], [0, lag]) for lag in (0, 10, 20, 30)}
    print("%s: N = 200 on one stream %.1f us per pass; one half alone %.1f; two halves on two streams, second lagging 0 / 10 / 20 / 30 us: %s"
          % (name, t_full, t_half, " / ".join("%.1f" % (res[k] - k / reps) for k in (0, 10, 20, 30))))
