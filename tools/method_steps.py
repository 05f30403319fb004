#!/usr/bin/env python
"""One method's training / importance batch in a short loop (for rocprofv3 --kernel-trace --stats): the same calls as
bench.py's `configs`.  usage: method_steps.py <hat|packnet|mas|si> [hw] [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clsurvey_amd import models, net, ops
which = sys.argv[1]
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda")
N = 200 if hw == 64 else 50
g = torch.Generator(device=dev); g.manual_seed(11)
x = torch.randn((N, 3, hw, hw), generator=g, device=dev)
y = torch.randint(0, 20, (N,), generator=g, device=dev)
if which in ("mas", "si"):
    m = models.parse_model_name("base_VGG9_cl_512_512", (hw, hw), 20)
    eng = net.NetEngine(m, N, (3, hw, hw), dev)
    A = eng.arena
    omega, init_val, w, buf = A.buffer("omega"), A.buffer("init_val"), A.buffer("w"), A.buffer("buf")
    init_val.copy_(A.theta)
    it = [0]
    if which == "mas":
        def step():
            eng.loss_step(x, None, "mse_sum_zero", True)
            ops.mas_accum(omega, A.grad, it[0], N); it[0] += 1
    else:
        def step():
            eng.loss_step(x, y, "ce_mean", True)
            ops.si_step(A.theta, A.grad, omega, init_val, w, buf, 400.0, 1e-3, 0.9, 0.0, it[0] == 0); it[0] += 1
else:
    m = models.parse_model_name("wide_VGG9_cl_512_512", (hw, hw), 20)
    if which == "packnet":
        from clsurvey_amd.methods import packnet as PK
        eng = net.NetEngine(m, N, (3, hw, hw), dev)
        A = eng.arena
        buf = torch.zeros_like(A.theta)
        mask = torch.randint(1, 3, (A.numel,), generator=g, device=dev, dtype=torch.int64).to(torch.uint8)
        first = [True]
        def step():
            eng.loss_step(x, y, "ce_mean", True)
            PK.fused_batch_tail(A.theta, A.grad, buf, mask, 2, 1e-3, 0.9, 0.0, first[0]); first[0] = False
    else:
        from clsurvey_amd.methods import hat as H
        hn = H.HatNet(m, (3, hw, hw), [(0, 20), (1, 20)]).to(dev)
        hat = H.HatEngine(hn, N, (3, hw, hw), dev)
        mask_pre, mask_back = H.init_masks(hat, 1, 800.0)
        opt = H.HAT_SGD(hn.parameters(), lr=1e-3, momentum=0.9, weight_decay=0.0)
        count = float(sum(float((1 - mp).sum().item()) for mp in mask_pre))
        def step():
            hat.step(1, x, y, 400.0, mask_pre, 2.5, count, backward=True)
            opt.step(hn, mask_back, 1, 400.0, 50, 800.0, 10000, thres_emb=6.0)
for _ in range(2):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print("%s hw=%d N=%d: %.3f ms/step" % (which, hw, N, (time.perf_counter() - t0) / steps * 1e3))
