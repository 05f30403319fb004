#!/usr/bin/env python
"""Does the chip fill one pass's launch tails with another pass's blocks?  small_VGG9 train pass at N = 200 on one stream vs
two independent half-batch passes (N = 100 each) on two streams vs the same two halves back to back on one stream.
usage: two_stream_probe.py [model] [N]"""
import copy, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clsurvey_amd import models
from clsurvey_amd.net import NetEngine
name = sys.argv[1] if len(sys.argv) > 1 else "small_VGG9_cl_128_128"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda")
m = models.parse_model_name(name, (64, 64), 20)
full = NetEngine(m, N, (3, 64, 64), dev)
x = torch.randn(N, 3, 64, 64, device=dev); y = torch.randint(0, 20, (N,), device=dev)


def timed(fn, it=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e6


print("one stream, N=%d: %.1f us / pass" % (N, timed(lambda: full.loss_step(x, y, "ce_mean", True))))
for parts in (2, 3, 4):
    n = N // parts
    engs = [NetEngine(copy.deepcopy(m), n, (3, 64, 64), dev) for _ in range(parts)]
    xs = [x[i * n:(i + 1) * n].contiguous() for i in range(parts)]
    ys = [y[i * n:(i + 1) * n].contiguous() for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    main = torch.cuda.current_stream()

    def serial():
        for e, a, b in zip(engs, xs, ys):
            e.loss_step(a, b, "ce_mean", True)

    def forked():
        ev = torch.cuda.Event(); ev.record(main)
        for s, e, a, b in zip(streams, engs, xs, ys):
            s.wait_event(ev)
            with torch.cuda.stream(s):
                e.loss_step(a, b, "ce_mean", True)
            d = torch.cuda.Event(); d.record(s); main.wait_event(d)
    print("%d parts of N=%d: back to back %.1f us, on %d streams %.1f us" % (parts, n, timed(serial), parts, timed(forked)))
