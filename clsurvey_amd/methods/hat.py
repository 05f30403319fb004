"""HAT on the HIP path — mirror of src/methods/HAT/networks/vgg_hat.py (Net), approaches/hat.py (Appr:
init_masks, criterion, train_epoch step) and HAT_utils.py (HAT_SGD).

The per-channel gates are folded into the NEXT layer's weights (see csrc/hat.hip), so the un-gated
activations flow through the ordinary NetEngine plan; only parameter-sized kernels are HAT specific.
"""
import copy

import torch
import torch.nn as nn

from .. import _lib
from .._lib import check
from ..net import NetEngine


def _stream():
    return torch.cuda.current_stream().cuda_stream


class HatNet(nn.Module):
    """Same parameter tree as vgg_hat.Net (vgg_hat.py:14-81): convs, conv_embs, fcs, fc_embs, classifier."""

    first_drop = False           # alexnet_hat.Net: relu(fc(drop(x))) instead of drop(relu(fc(x))) (vgg_hat.py:110-114)

    def __init__(self, rawmodel, inputsize, taskcla, uniform_init=True):
        super().__init__()
        self.taskcla = taskcla
        self.pool_geometry = None    # (kernel, stride) of the raw model's first MaxPool2d: one pool module serves all (:41-44)
        self.drop_p = 0.0            # p of the raw classifier's first Dropout; 0 = none (:47, :61-62)
        self.convs, self.conv_embs = nn.ModuleList(), nn.ModuleList()
        self.fcs, self.fc_embs = nn.ModuleList(), nn.ModuleList()
        self.classifier = nn.ModuleList()
        self.maxpool_idxs = []
        conv_idx = 0
        for mod in rawmodel.features.children():
            if isinstance(mod, nn.Conv2d):
                self.convs.append(copy.deepcopy(mod))
                self.conv_embs.append(nn.Embedding(len(taskcla), mod.out_channels))
                conv_idx += 1
            elif isinstance(mod, nn.MaxPool2d):
                if self.pool_geometry is None:
                    one = lambda v: v if isinstance(v, int) else v[0]
                    self.pool_geometry = (one(mod.kernel_size), one(mod.stride))
                self.maxpool_idxs.append(conv_idx - 1)
        for mod in rawmodel.classifier.children():
            if isinstance(mod, nn.Dropout) and self.drop_p == 0.0:
                self.drop_p = float(mod.p)
        fcs = [m for m in rawmodel.classifier.children() if isinstance(m, nn.Linear)]
        for i, mod in enumerate(fcs):
            if i < len(fcs) - 1:
                self.fcs.append(copy.deepcopy(mod))
                self.fc_embs.append(nn.Embedding(len(taskcla), mod.out_features))
            else:
                self.classifier.append(copy.deepcopy(mod))
        smid_sq = self.fcs[0].in_features / self.convs[-1].out_channels
        self.smid = int(smid_sq ** 0.5)
        assert self.convs[-1].out_channels * self.smid * self.smid == self.fcs[0].in_features
        self.enable_warmup = True
        self.smax = None
        self.lamb = None
        if uniform_init:
            for emb in list(self.conv_embs) + list(self.fc_embs):
                emb.weight.data.uniform_(0, 2)

    def plain_view(self):
        """features / classifier Sequentials over the SAME Conv2d / Linear modules (for NetEngine)."""
        feats = []
        pk, ps = getattr(self, "pool_geometry", None) or (2, 2)
        for i, c in enumerate(self.convs):
            feats += [c, nn.ReLU(inplace=True)]
            if i in self.maxpool_idxs:
                feats.append(nn.MaxPool2d(pk, ps))
        cls = []
        drop_p = getattr(self, "drop_p", 0.0)
        for f in self.fcs:
            if drop_p > 0 and self.first_drop:
                cls.append(nn.Dropout(drop_p))
            cls += [f, nn.ReLU(True)]
            if drop_p > 0 and not self.first_drop:
                cls.append(nn.Dropout(drop_p))
        cls.append(self.classifier[0])
        view = nn.Module()
        view.features = nn.Sequential(*feats)
        view.classifier = nn.Sequential(*cls)
        return view


class HatNetAlexnet(HatNet):
    """alexnet_hat.Net (networks/alexnet_hat.py:4-13): the same dynamic construction over torchvision's AlexNet tree, dropout
    IN FRONT of each gated Linear layer, no warm-up, a 6x6 feature map behind the convolutions."""
    first_drop = True

    def __init__(self, *args, **kwargs):
        kwargs["uniform_init"] = True
        super().__init__(*args, **kwargs)
        self.enable_warmup = False
        assert self.smid == 6


class HatEngine:
    """forward / criterion / backward of vgg_hat.Net.forward + Appr.criterion for one batch."""

    def __init__(self, net, max_batch, in_shape, device="cuda"):
        self.net = net.to(device)
        self.device = torch.device(device)
        self.view = net.plain_view()
        self.engine = NetEngine(self.view, max_batch, in_shape, device)
        self.A = self.engine.arena
        self.scaled = torch.zeros_like(self.A.theta)
        self.layers = list(net.convs) + list(net.fcs) + [net.classifier[0]]          # in plan order
        self.embs = list(net.conv_embs) + list(net.fc_embs)                           # gate l follows layer l
        self.nc = len(net.convs)
        self.gate = [torch.zeros(e.weight.shape[1], device=self.device) for e in self.embs]
        self.dgate = [torch.zeros_like(g) for g in self.gate]
        self.reg_sums = torch.zeros(3, dtype=torch.float64, device=self.device)      # sum a(1-m), sum (1-m), their ratio
        for e in self.embs:
            e.weight.grad = torch.zeros_like(e.weight.data)
        # static job tables of the whole-net launches (every pointer below is fixed for the engine's lifetime)
        jobs = []
        for li, mod in enumerate(self.layers):
            w, b = mod.weight, mod.bias
            ow, nw = self.A.slot(w)
            gin = self.gate[li - 1] if li > 0 else None
            R = self._R(li)
            Cg = gin.numel() if gin is not None else nw // (w.shape[0] * R)
            jobs.append(_lib.HatLayer(self.A.theta[ow:ow + nw].data_ptr(), gin.data_ptr() if gin is not None else None,
                                      self.scaled[ow:ow + nw].data_ptr(), w.shape[0], Cg, R))
            ob, nbias = self.A.slot(b)
            jobs.append(_lib.HatLayer(self.A.theta[ob:ob + nbias].data_ptr(), None, self.scaled[ob:ob + nbias].data_ptr(), nbias, 1, 1))
        self._scale_jobs = (_lib.HatLayer * len(jobs))(*jobs)
        wg = []
        for li in range(1, len(self.layers)):
            w = self.layers[li].weight
            ow, nw = self.A.slot(w)
            gin = self.gate[li - 1]
            wg.append(_lib.HatWgradJob(self.A.grad[ow:ow + nw].data_ptr(), self.A.theta[ow:ow + nw].data_ptr(), gin.data_ptr(),
                                       self.dgate[li - 1].data_ptr(), w.shape[0], gin.numel(), self._R(li), 0))
        self._wgrad_jobs = (_lib.HatWgradJob * len(wg))(*wg)

    def _R(self, li):
        """inner repeat of layer li's input-channel index in its weight layout [K][C][R]."""
        if li < self.nc:
            w = self.layers[li].weight
            return int(w.shape[2] * w.shape[3])
        if li == self.nc:
            return self.net.smid * self.net.smid
        return 1

    def gates(self, t, s, mask_pre=None, sums=None):
        """gate_l = sigmoid(s * E_l[t]) for every layer in ONE launch (vgg_hat.py:121-127); with `sums` (2 doubles on the
        device) also the regulariser's sum gate * (1 - mask_pre) and sum (1 - mask_pre) (hat.py:285-299)."""
        jobs = (_lib.HatGateJob * len(self.embs))()
        for l, (e, g) in enumerate(zip(self.embs, self.gate)):
            n = g.numel()
            jobs[l] = _lib.HatGateJob(e.weight.data.data_ptr() + 4 * n * int(t), g.data_ptr(),
                                      mask_pre[l].data_ptr() if mask_pre is not None else None, n, 0)
        check(_lib.lib().clhip_hat_gates_multi(jobs, len(self.embs), float(s), sums.data_ptr() if sums is not None else None,
                                               _stream()), "clhip_hat_gates_multi")
        return self.gate

    def masks_at(self, t, s):
        return [g.clone() for g in self.gates(t, s)]

    def _scale_weights(self):
        """W' = W * gate_in[c] for every layer, biases and the first layer copied: ONE launch over the arena."""
        check(_lib.lib().clhip_hat_scale_weights_multi(self._scale_jobs, len(self._scale_jobs), _stream()),
              "clhip_hat_scale_weights_multi")

    def forward(self, t, x, s):
        """vgg_hat.Net.forward logits for task t at gate slope s (inference: s = smax)."""
        self.gates(t, s)
        self._scale_weights()
        return self.engine.forward(x, params=self.scaled)

    def plain_step(self, x, y, backward=True, stats=None):
        """hat_finetune.py:133-137: forward with every gate forced to 1 (masks=ft_mask) + CE (+ backward)."""
        return self.engine.loss_step(x, y, "ce_mean", backward, stats)

    def step(self, t, x, y, s, mask_pre=None, lamb=0.0, count=None, backward=True, stats=None, want_logits=False):
        """Returns (ce_loss[1] device, reg device scalar (lamb*reg), logits|None). With backward=True all
        .grad fields (convs, fcs, head in the arena; embeddings dense with row t filled) are set.  Six launches next to
        the net's own plan, no host synchronisation (count = None reads the regulariser's denominator on the device)."""
        L = _lib.lib()
        self.gates(t, s, mask_pre, self.reg_sums)
        self._scale_weights()
        ce, logits = self.engine.loss_step(x, y, "ce_mean", backward, stats, want_logits, params=self.scaled)
        reg = lamb * self.reg_sums[2]
        if backward:
            check(L.clhip_hat_weight_grads_multi(self._wgrad_jobs, len(self._wgrad_jobs), _stream()), "clhip_hat_weight_grads_multi")
            jobs = (_lib.HatEmbJob * len(self.embs))()
            for l, e in enumerate(self.embs):
                rows, n = e.weight.shape
                jobs[l] = _lib.HatEmbJob(self.dgate[l].data_ptr(), self.gate[l].data_ptr(),
                                         mask_pre[l].data_ptr() if mask_pre is not None else None, e.weight.grad.data_ptr(),
                                         n, rows, int(t), 0)
            check(L.clhip_hat_emb_grads_multi(jobs, len(self.embs), float(s), float(lamb), float(count) if count else 0.0,
                                              self.reg_sums.data_ptr(), _stream()), "clhip_hat_emb_grads_multi")
        return ce, reg, logits


def init_masks(hat, current_task, smax):
    """hat.py:58-89 on the device: (mask_pre list, mask_back {param name: tensor})."""
    L = _lib.lib()
    mask_pre = None
    for t in range(current_task):
        m = hat.masks_at(t, smax)
        mask_pre = m if mask_pre is None else [torch.max(a, b) for a, b in zip(mask_pre, m)]
    mask_back = {}
    if mask_pre is None:
        return None, mask_back
    net, nc = hat.net, hat.nc

    def bm(post, pre, shape, C, R):
        out = torch.empty(shape, dtype=torch.float32, device=hat.device)
        check(L.clhip_hat_backmask(post.data_ptr(), pre.data_ptr() if pre is not None else None, out.data_ptr(),
                                   shape[0], C, R, _stream()), "clhip_hat_backmask")
        return out
    for i, conv in enumerate(net.convs):
        pre = mask_pre[i - 1] if i > 0 else None
        mask_back["convs.%d.weight" % i] = bm(mask_pre[i], pre, tuple(conv.weight.shape), conv.in_channels,
                                              int(conv.weight.shape[2] * conv.weight.shape[3]))
        mask_back["convs.%d.bias" % i] = bm(mask_pre[i], None, tuple(conv.bias.shape), 1, 1)      # :275-276
    for i, fc in enumerate(net.fcs):
        pre = mask_pre[nc + i - 1] if i > 0 else mask_pre[nc - 1]
        R = net.smid * net.smid if i == 0 else 1
        mask_back["fcs.%d.weight" % i] = bm(mask_pre[nc + i], pre, tuple(fc.weight.shape), pre.numel(), R)
        mask_back["fcs.%d.bias" % i] = bm(mask_pre[nc + i], None, tuple(fc.bias.shape), 1, 1)     # :292-293
    return mask_pre, mask_back


class HAT_SGD(torch.optim.Optimizer):
    """HAT_utils.py:185-250."""

    def __init__(self, params, lr, momentum=0, dampening=0, weight_decay=0, nesterov=False):
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                                      nesterov=nesterov))
        self._ws = None

    def step(self, model, mask_back, t, s=None, thres_cosh=None, smax=None, clipgrad=None, finetune=False, closure=None,
             thres_emb=None):
        """HAT_utils.py:192-250 for every parameter in two launches (clip_grad_norm_ stays per parameter); thres_emb also
        applies the embedding clamp that follows optimizer.step in the reference's batch loop (hat.py:238-240)."""
        L = _lib.lib()
        for group in self.param_groups:
            rows, fresh = [], []
            for p, (name, modp) in zip(group["params"], model.named_parameters()):
                assert modp is p
                if p.grad is None:
                    continue
                st = self.state[p]
                first = "momentum_buffer" not in st
                if first:
                    st["momentum_buffer"] = torch.zeros_like(p.data)
                fresh.append(first)
                mb = mask_back.get(name) if t > 0 else None
                rows.append(_lib.HatParam(p.data.data_ptr(), p.grad.data.data_ptr(), st["momentum_buffer"].data_ptr(),
                                          mb.data_ptr() if mb is not None else None, p.numel(), int("embs" in name), 0))
            if not rows:
                continue
            if any(fresh) and not all(fresh):
                raise RuntimeError("HAT_SGD: parameters joined the optimizer after its first step")
            if self._ws is None or self._ws.numel() < L.clhip_hat_sgd_multi_ws(len(rows)):
                self._ws = torch.zeros(L.clhip_hat_sgd_multi_ws(len(rows)), dtype=torch.uint8, device=group["params"][0].device)
            table = (_lib.HatParam * len(rows))(*rows)
            check(L.clhip_hat_sgd_step_multi(table, len(rows), float(group["lr"]), float(group["momentum"]),
                                             float(group["weight_decay"]), int(finetune), float(s or 0.0), float(smax or 0.0),
                                             float(thres_cosh or 0.0), float(clipgrad or 0.0), float(thres_emb or 0.0),
                                             int(fresh[0]), self._ws.data_ptr(), self._ws.numel(), _stream()),
                  "clhip_hat_sgd_step_multi")
        return None


def clamp_embeddings(net, thres_emb=6.0):
    """hat.py:238-240."""
    L = _lib.lib()
    for n, p in net.named_parameters():
        if "embs" in n:
            check(L.clhip_clamp(p.data.data_ptr(), p.numel(), -thres_emb, thres_emb, _stream()), "clhip_clamp")
