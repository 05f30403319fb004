"""ParamArena + NetEngine: the MI355X execution model for a VGG-style net.

* ParamArena lays all parameters of a model out in ONE contiguous fp32 HBM buffer (16-byte
  aligned slots, module order) and re-points every nn.Parameter's .data / .grad at views of it.
  The Parameter OBJECTS are untouched, so the reference's `reg_params` dict (keyed by Parameter
  identity, pickled with the model: EWC/main_EWC.py:160-232) keeps working.  Every optimizer /
  importance update then is a single HBM-bound kernel over the arena instead of ~8 eager ops
  x 16 tensors (EWC/train_EWC.py:46-84).
* NetEngine parses model.features / model.classifier (the structure of models/VGGSlim.py:27-76)
  into a static plan executed by libclhip's clhip_net_* entry points: one call per pass.
"""
import ctypes as C
import weakref

import torch
import torch.nn as nn

from . import _lib
from ._lib import check, LayerDesc


def _pad4(n):
    return (n + 3) // 4 * 4


class ParamArena:
    _registry = weakref.WeakValueDictionary()   # id(first param) -> arena

    @classmethod
    def find(cls, params):
        """The arena that owns exactly this parameter list (or None)."""
        params = list(params)
        if not params:
            return None
        a = cls._registry.get(id(params[0]))
        if a is None or len(a.params) != len(params) or any(x is not y for x, y in zip(a.params, params)):
            return None
        return a

    def __init__(self, params, device=None):
        self.params = list(params)
        if device is None:
            device = self.params[0].device
        self.device = torch.device(device)
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += _pad4(p.numel())
        self.numel = off
        self.theta = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.aux = {}
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                v = self.theta[o:o + p.numel()].view(p.shape)
                v.copy_(p.data.to(self.device, torch.float32))
                p.data = v
                p.grad = self.grad[o:o + p.numel()].view(p.shape)
        self._index = {id(p): i for i, p in enumerate(self.params)}
        ParamArena._registry[id(self.params[0])] = self

    def slot(self, p):
        i = self._index[id(p)]
        return self.offsets[i], p.numel()

    def buffer(self, name, zero=True):
        """Named auxiliary arena (omega, init_val, momentum buffer, w, ...)."""
        if name not in self.aux:
            self.aux[name] = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        elif zero:
            pass
        return self.aux[name]

    def view(self, name, p):
        o, n = self.slot(p)
        buf = self.theta if name == "theta" else self.grad if name == "grad" else self.buffer(name, zero=False)
        return buf[o:o + n].view(p.shape)

    def load(self, name, per_param):
        """Fill aux arena `name` from {Parameter: tensor}; params missing from the dict get zeros
        (e.g. omega of a fresh head that is 'not in reg_params')."""
        buf = self.buffer(name)
        buf.zero_()
        for p in self.params:
            t = per_param.get(p)
            if t is not None:
                self.view(name, p).copy_(t.to(self.device, torch.float32))
        return buf

    def zero_grad(self):
        self.grad.zero_()


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def parse_net(model, with_bn=False):
    """([(kind, module, relu, pool)], {plan index: nn.Dropout in front of that layer}) from a module with the
    features / classifier structure of models/VGGSlim.py:27-76 or of torchvision's AlexNet (models/net.py:96-125).
    pool is False, True (2x2 stride 2) or (k, stride).  Raises on anything the static plan does not cover (BatchNorm
    variants: only with with_bn=True, which appends {plan index: nn.BatchNorm2d behind that convolution} to the result)."""
    layers, drops, bns = [], {}, {}
    feats = list(model.features.children())
    i = 0
    pending = None
    while i < len(feats):
        m = feats[i]
        if isinstance(m, nn.Dropout):
            pending = m
            i += 1
        elif isinstance(m, nn.Conv2d):
            ks, st, pd = _pair(m.kernel_size), _pair(m.stride), _pair(m.padding)
            if ks[0] != ks[1] or st[0] != st[1] or pd[0] != pd[1] or m.groups != 1 or _pair(m.dilation) != (1, 1):
                raise NotImplementedError("NetEngine: square kernels, symmetric stride / padding, no groups / dilation")
            j = i + 1
            if j < len(feats) and isinstance(feats[j], nn.BatchNorm2d):
                bn = feats[j]
                if not with_bn:
                    raise NotImplementedError("this caller does not handle BatchNorm2d layers")
                if not bn.affine or not bn.track_running_stats or bn.momentum is None or bn.num_features != m.out_channels:
                    raise NotImplementedError("NetEngine: BatchNorm2d must be affine with running statistics and a momentum")
                bns[len(layers)] = bn
                j += 1
            relu = j < len(feats) and isinstance(feats[j], nn.ReLU)
            j += 1 if relu else 0
            pool = False
            if j < len(feats) and isinstance(feats[j], nn.MaxPool2d):
                mp = feats[j]
                pk, ps = _pair(mp.kernel_size), _pair(mp.stride if mp.stride is not None else mp.kernel_size)
                if pk[0] != pk[1] or ps[0] != ps[1] or _pair(mp.padding) != (0, 0) or mp.ceil_mode or _pair(mp.dilation) != (1, 1):
                    raise NotImplementedError("NetEngine: square un-padded floor-mode max-pool only")
                pool = True if (pk[0], ps[0]) == (2, 2) else (pk[0], ps[0])
                j += 1
            if pending is not None:
                drops[len(layers)] = pending
                pending = None
            layers.append(("conv", m, relu, pool))
            i = j
        else:
            raise NotImplementedError("NetEngine: unsupported feature module %r" % (m,))
    cls = list(model.classifier.children())
    i = 0
    while i < len(cls):
        m = cls[i]
        if isinstance(m, nn.Dropout):
            pending = m
            i += 1
        elif isinstance(m, nn.Linear):
            relu = i + 1 < len(cls) and isinstance(cls[i + 1], nn.ReLU)
            if pending is not None:
                drops[len(layers)] = pending
                pending = None
            layers.append(("fc", m, relu, False))
            i += 2 if relu else 1
        else:
            raise NotImplementedError("NetEngine: unsupported classifier module %r" % (m,))
    if pending is not None or 0 in drops:
        raise NotImplementedError("NetEngine: Dropout must sit in front of a layer other than the first")
    return (layers, drops, bns) if with_bn else (layers, drops)


def parse_vgg(model):
    """The plan layers only (see parse_net)."""
    return parse_net(model)[0]


def conv_geometry(m):
    """(ksize, stride, pad) of a Conv2d accepted by parse_net."""
    return _pair(m.kernel_size)[0], _pair(m.stride)[0], _pair(m.padding)[0]


class NetEngine:
    LOSS = {"ce_mean": 0, "ce_sum": 1, "mse_sum_zero": 2}

    def __init__(self, model, max_batch, in_shape, device="cuda", layers=None, params=None, drops=None):
        """layers / params (optional): an explicit plan [(kind, weight, bias, cin, cout, relu, pool)] and the parameter
        order of the arena, for models whose module tree is not plain VGGSlim (e.g. LwF's stacked heads run as ONE
        Linear over head parameters laid out back to back)."""
        self.model = model
        self.device = torch.device(device)
        self.drops, self.bns = {}, {}
        if layers is None:
            self.layers, self.drops, self.bns = parse_net(model, with_bn=True)
            specs = [(kind, m.weight, m.bias, m.in_channels if kind == "conv" else m.in_features,
                      m.out_channels if kind == "conv" else m.out_features, relu, pool) +
                     ((conv_geometry(m),) if kind == "conv" else ()) for kind, m, relu, pool in self.layers]
        else:
            specs = list(layers)
            self.layers = specs
            self.drops = dict(drops or {})
        if any(sp[2] is None for sp in specs):
            raise NotImplementedError("NetEngine: layers without bias")
        self.arena = ParamArena(list(model.parameters()) if params is None else list(params), self.device)
        descs = (LayerDesc * len(specs))()
        for d, sp in zip(descs, specs):
            kind, w, b, cin, cout, relu, pool = sp[:7]
            d.type = 0 if kind == "conv" else 1
            d.cin, d.cout = int(cin), int(cout)
            d.relu, d.pool = int(relu), int(bool(pool))
            if isinstance(pool, tuple):
                d.pool_k, d.pool_s = int(pool[0]), int(pool[1])
            if kind == "conv":
                d.ksize, d.stride, d.pad = (int(v) for v in (sp[7] if len(sp) > 7 else (w.shape[2], 1, w.shape[2] // 2)))
            d.w_off = self.arena.slot(w)[0]
            d.b_off = self.arena.slot(b)[0]
        for li in self.drops:
            descs[li].has_drop = 1
        for li, bn in self.bns.items():
            descs[li].bn = 1
            descs[li].bn_w_off = self.arena.slot(bn.weight)[0]
            descs[li].bn_b_off = self.arena.slot(bn.bias)[0]
        self.max_batch = int(max_batch)
        self.in_shape = tuple(in_shape)
        h = C.c_void_p()
        L = _lib.lib()
        check(L.clhip_net_create(descs, len(specs), self.max_batch, *self.in_shape, C.byref(h)),
              "clhip_net_create")
        self._h = h
        self.n_classes = L.clhip_net_num_classes(h)
        self.ws = torch.empty(L.clhip_net_workspace_bytes(h), dtype=torch.uint8, device=self.device)
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        for li, bn in self.bns.items():      # running statistics stay module buffers (pickled with the model), on the device
            for name in ("running_mean", "running_var", "num_batches_tracked"):
                buf = getattr(bn, name)
                buf.data = buf.data.to(self.device).contiguous()
            check(L.clhip_net_set_bn(h, li, bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(bn.momentum), float(bn.eps)),
                  "clhip_net_set_bn")
        self._training = None
        self.in_elems = {}
        shp = self.in_shape
        if self.drops:
            with torch.no_grad():       # input width of every layer with a Dropout in front (shape arithmetic only)
                c, h, w = shp
                for li, sp in enumerate(specs):
                    self.in_elems[li] = c * h * w
                    if sp[0] == "conv":
                        ks, st, pd = sp[7] if len(sp) > 7 else (3, 1, 1)
                        c, h, w = sp[4], (h + 2 * pd - ks) // st + 1, (w + 2 * pd - ks) // st + 1
                        if sp[6]:
                            pk, ps = sp[6] if isinstance(sp[6], tuple) else (2, 2)
                            h, w = (h - pk) // ps + 1, (w - pk) // ps + 1
                    else:
                        c, h, w = sp[4], 1, 1
        self.auto_dropout = True     # loss_step / forward draw nn.Dropout masks themselves while model.training
        self._masks = {}

    def set_dropout(self, layer, mask):
        """mask: None (off), [in_elems] (one row for the whole batch: GEM) or [N][in_elems] device fp32, values 0 or
        1/p_retain, multiplying the input of plan layer `layer`.  Stays in force until changed."""
        if mask is None:
            self._masks.pop(layer, None)
            ptr, stride = None, 0
        else:
            if not mask.is_cuda or mask.dtype != torch.float32 or not mask.is_contiguous():
                raise RuntimeError("dropout masks are contiguous fp32 HIP tensors")
            if mask.shape[-1] != self.in_elems[layer] or (mask.dim() == 2 and mask.shape[0] < 1):
                raise RuntimeError("dropout mask shape %s does not fit layer %d" % (tuple(mask.shape), layer))
            self._masks[layer] = mask
            ptr, stride = mask.data_ptr(), (mask.shape[-1] if mask.dim() == 2 else 0)
        check(_lib.lib().clhip_net_set_dropout(self._h, int(layer), ptr, int(stride)), "clhip_net_set_dropout")

    def layer_input(self, layer, n):
        """[n][in_elems] view (no copy) of the activation that feeds plan layer `layer` (> 0), valid after a forward."""
        off, elems = C.c_size_t(), C.c_size_t()
        check(_lib.lib().clhip_net_layer_input(self._h, int(layer), C.byref(off), C.byref(elems)), "clhip_net_layer_input")
        lo = off.value * 4
        return self.ws[lo:lo + n * elems.value * 4].view(torch.float32).view(n, elems.value)

    def pool_idx(self, layer, n):
        """[n][elems] uint8 view of the arg-max codes (window position r*k + c) the last forward stored for the
        max-pooled conv layer `layer`.  The fused conv + ReLU + 2x2-pool kernels write 4 (no position) for a window whose
        maximum after ReLU is not positive: no gradient passes through it (csrc/common.hpp, CLHIP_POOL_DEAD)."""
        off, elems = C.c_size_t(), C.c_size_t()
        check(_lib.lib().clhip_net_layer_pool_idx(self._h, int(layer), C.byref(off), C.byref(elems)), "clhip_net_layer_pool_idx")
        return self.ws[off.value:off.value + n * elems.value].view(n, elems.value)

    def layer_paths(self, layer):
        """{'fwd', 'bwd_data', 'bwd_weight'} -> True where the plan runs the layer's kernel through a prepared-weights path
        (Winograd, csrc/wino.hip) instead of the direct f32 kernels; 'bs_fwd' / 'bs_bwd_data' -> True where that launch is the
        bf16-split kernel (csrc/bsconv.hip) instead of Winograd; 'bs_bwd_weight' -> the weight gradient is csrc/bswgrad.hip."""
        bits = _lib.lib().clhip_net_layer_paths(self._h, int(layer))
        if bits < 0:
            raise RuntimeError("clhip_net_layer_paths(%d)" % layer)
        return {"fwd": bool(bits & 1), "bwd_data": bool(bits & 2), "bwd_weight": bool(bits & 4),
                "bs_fwd": bool(bits & 8), "bs_bwd_data": bool(bits & 16), "bs_bwd_weight": bool(bits & 32)}

    def set_input_grad(self, layer, extra):
        """extra [N][in_elems] (or None) is added to the gradient w.r.t. layer_input(layer) in the following backward
        passes; if that activation is a ReLU output the caller masks extra with (activation > 0) first."""
        if extra is not None and (not extra.is_cuda or extra.dtype != torch.float32 or not extra.is_contiguous()):
            raise RuntimeError("set_input_grad needs a contiguous fp32 HIP tensor")
        self._extra = extra          # keep it alive
        check(_lib.lib().clhip_net_set_input_grad(self._h, int(layer), extra.data_ptr() if extra is not None else None),
              "clhip_net_set_input_grad")

    def _mode(self):
        """nn.Module.train / eval -> the plan's BatchNorm mode; num_batches_tracked as nn.BatchNorm2d.forward counts it."""
        if not self.bns:
            return
        training = bool(self.model.training)
        if training != self._training:
            check(_lib.lib().clhip_net_set_training(self._h, int(training)), "clhip_net_set_training")
            self._training = training
        if training:
            for bn in self.bns.values():
                bn.num_batches_tracked += 1

    def probe(self, layer, kind="fwd"):
        """Measurement: HIP events around the forward ('fwd'), backward-data ('bwd_data') or weight-gradient ('bwd_weight')
        launch(es) of plan layer `layer` from now on (None: off)."""
        k = {"fwd": 0, "bwd_data": 1, "bwd_weight": 2}[kind]
        check(_lib.lib().clhip_net_probe_kind(self._h, -1 if layer is None else int(layer), k), "clhip_net_probe_kind")

    def probe_read(self):
        """(average microseconds, passes covered) of the probed layer's forward launch since the last read (at most 64)."""
        import ctypes as C
        us, n = C.c_float(0.0), C.c_int(0)
        check(_lib.lib().clhip_net_probe_read(self._h, C.byref(us), C.byref(n)), "clhip_net_probe_read")
        return float(us.value), int(n.value)

    def _auto_drop(self, n):
        """nn.Dropout semantics (fresh Bernoulli(1-p)/(1-p) mask per element per pass while model.training, identity in
        eval mode) for the Dropout modules of the plan; drawn with torch's device generator."""
        if not self.drops or not self.auto_dropout:
            return
        for li, m in self.drops.items():
            if self.model.training and m.p > 0:
                keep = 1.0 - m.p
                mask = torch.empty((n, self.in_elems[li]), dtype=torch.float32, device=self.device).bernoulli_(keep).div_(keep)
                self.set_dropout(li, mask)
            elif li in self._masks:
                self.set_dropout(li, None)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().clhip_net_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _check_x(self, x):
        if not x.is_cuda or x.dtype != torch.float32 or not x.is_contiguous():
            raise RuntimeError("NetEngine needs contiguous fp32 HIP tensors")
        if tuple(x.shape[1:]) != self.in_shape or x.shape[0] > self.max_batch:
            raise RuntimeError("NetEngine: bad input shape %s" % (tuple(x.shape),))

    def forward(self, x, params=None):
        self._check_x(x)
        self._mode()
        self._auto_drop(x.shape[0])
        logits = torch.empty((x.shape[0], self.n_classes), dtype=torch.float32, device=self.device)
        check(_lib.lib().clhip_net_forward(self._h, (params if params is not None else self.arena.theta).data_ptr(),
                                           x.data_ptr(), x.shape[0],
                                           self.ws.data_ptr(), logits.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream), "clhip_net_forward")
        return logits

    def backward(self, x, dlogits):
        """Backward of the last forward() on this engine from dlogits [N][classes] into arena.grad."""
        self._check_x(x)
        check(_lib.lib().clhip_net_backward(self._h, self.arena.theta.data_ptr(), self.arena.grad.data_ptr(), x.data_ptr(),
                                            x.shape[0], self.ws.data_ptr(), dlogits.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream), "clhip_net_backward")

    def loss_step(self, x, y, kind="ce_mean", backward=True, stats=None, want_logits=False, params=None,
                  class_slice=None):
        """forward + loss (+ backward into arena.grad). Returns (loss[1] device tensor, logits|None).
        No host synchronisation happens here."""
        self._check_x(x)
        self._mode()
        self._auto_drop(x.shape[0])
        logits = torch.empty((x.shape[0], self.n_classes), dtype=torch.float32, device=self.device) if want_logits else None
        o1, nc = (class_slice[0], class_slice[1] - class_slice[0]) if class_slice is not None else (0, 0)
        check(_lib.lib().clhip_net_loss_step_slice(
            self._h, (params if params is not None else self.arena.theta).data_ptr(),
            self.arena.grad.data_ptr() if backward else None, x.data_ptr(),
            y.data_ptr() if y is not None else None, x.shape[0], self.LOSS[kind], int(o1), int(nc), self.ws.data_ptr(),
            self.loss.data_ptr(), stats.data_ptr() if stats is not None else None,
            logits.data_ptr() if logits is not None else None, torch.cuda.current_stream().cuda_stream),
            "clhip_net_loss_step_slice")
        return self.loss, logits
