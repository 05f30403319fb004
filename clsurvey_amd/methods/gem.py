"""GEM on the HIP path — mirror of src/methods/rehearsal/model/gem.py (Net.observe / forward /
fill_buffer / manage_memory) with the exemplar ring buffer held as TENSORS in HBM.

The reference stores exemplar *paths* and re-decodes n_memories JPEGs for every past task on every
training batch (gem.py:233-235); here memory[t] is a device tensor and a past-task pass is
ceil(n_memories / batch) engine calls.  Gradients of a task are one contiguous row of G (the
ParamArena gradient is flat), the QP inputs come from ONE Gram-matrix pass and the tiny QP (quadprog's Goldfarb-Idnani)
runs on the device in float64 (clhip_gem_qp) — no host round trip per batch.
"""
import copy
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib
from .._lib import check
from ..data import DeviceLoader, TensorTaskDataset
from ..net import NetEngine
from ..optim import SGD


def _stream():
    return torch.cuda.current_stream().cuda_stream


def compute_offsets(task_idx, cum_nc_per_task):
    """rehearsal/model/common.py:106-118."""
    o1 = 0 if task_idx == 0 else int(cum_nc_per_task[task_idx - 1])
    return o1, int(cum_nc_per_task[task_idx])


def extend_head(model, n_outputs):
    """gem.py:99-113: replace the head by an n_outputs-way Linear whose first rows are the old head."""
    last = str(len(model.classifier._modules) - 1)
    old = copy.deepcopy(model.classifier._modules[last])
    new = nn.Linear(old.in_features, n_outputs)
    with torch.no_grad():
        new.weight[:old.out_features].copy_(old.weight)
        new.bias[:old.out_features].copy_(old.bias)
    model.classifier._modules[last] = new
    return model


class GemNet:
    """gem.Net (gem.py:83-387). Picklable like the reference's nn.Module (torch.save(model) /
    copy.deepcopy(model) in train_rehearsal.py:176-180): the pickle carries the wrapped net, the exemplar
    tensors and the counters; engine / workspaces are rebuilt on load (init_setup)."""

    def __init__(self, model, n_outputs, n_tasks, nc_per_task, n_memories, lr, weight_decay=0.0, memory_strength=1.0,
                 batch_size=200, in_shape=(3, 64, 64), device="cuda"):
        self.net = model.to(device)
        self.device = torch.device(device)
        self.n_outputs, self.n_tasks, self.n_memories = n_outputs, n_tasks, n_memories
        self.batch_size = batch_size
        self.in_shape = tuple(in_shape)
        self.memory_x = torch.zeros((n_tasks, n_memories) + self.in_shape, dtype=torch.float32, device=self.device)
        self.memory_labels = torch.zeros((n_tasks, n_memories), dtype=torch.int64, device=self.device)
        self.cum_nc_per_task = [sum(nc_per_task[:i + 1]) for i in range(len(nc_per_task))]
        self.observed_tasks, self.old_task, self.mem_cnt = [], -1, 0
        self._bind()
        self.init_setup(lr=lr, weight_decay=weight_decay, memory_strength=memory_strength)

    def _bind(self):
        self.engine = NetEngine(self.net, max(self.batch_size, 1), self.in_shape, self.device)
        self.engine.auto_dropout = False        # Dropout masks are GEM's own (below), not nn.Dropout's
        self.dropout_masks = {}
        self.A = self.engine.arena
        self.G = torch.zeros((self.n_tasks, self.A.numel), dtype=torch.float32, device=self.device)   # gem.py:131
        L = _lib.lib()
        self._gram_ws = torch.zeros(L.clhip_gem_gram_ws(16), dtype=torch.uint8, device=self.device)
        self._gram = torch.zeros(16 * 16, dtype=torch.float64, device=self.device)
        self._v = torch.zeros(16, dtype=torch.float64, device=self.device)          # QP solution, stays on the device
        self._info = torch.zeros(2, dtype=torch.int32, device=self.device)          # {violated constraints, status}
        self._qp_bad = torch.zeros(1, dtype=torch.int32, device=self.device)        # sticky: solves that did not report 'ok'
        self.host_qp = None          # tests only: a host solver f(gram, t, rows, margin) -> v replaces the device QP
        self.stats = torch.zeros(2, dtype=torch.float64, device=self.device)

    def init_setup(self, args=None, lr=None, weight_decay=None, memory_strength=None):
        """gem.py:146-155: fresh SGD(momentum 0.9) and margin; called after construction and after torch.load."""
        if args is not None:
            lr, weight_decay, memory_strength = args.lr, args.weight_decay, args.memory_strength
        self.dropout_masks = {}                                                                   # gem.py:152
        self.opt = SGD(self.net.parameters(), lr, momentum=0.9, weight_decay=weight_decay)       # gem.py:153
        self.margin = memory_strength

    # ------------------------------------------------------------------ gem.py:166-196 (manual dropout)
    def reset_dropout_config(self):
        self.dropout_masks = {}

    def _dropout(self, train, p_retain_unit=0.5):
        """Training mode: every Dropout of the plan multiplies its input by ONE mask row Bernoulli(p_retain)/p_retain of a
        single sample's shape, drawn when first needed after a reset and shared by all samples and passes until the next
        reset (gem.py:180-191; p_retain is the fixed 0.5 of the reference's signature, not module.p).  Eval: identity."""
        for li in self.engine.drops:
            if not train:
                self.engine.set_dropout(li, None)
                continue
            if li not in self.dropout_masks:
                self.dropout_masks[li] = self._draw_mask(li, self.engine.in_elems[li], p_retain_unit)
            self.engine.set_dropout(li, self.dropout_masks[li])

    def _draw_mask(self, layer, n, p_retain_unit):
        """gem.py:183-186: torch.bernoulli(fill(p_retain)) / p_retain over one sample's features (device generator)."""
        return torch.full((n,), p_retain_unit, dtype=torch.float32, device=self.device).bernoulli_().div_(p_retain_unit)

    _TRANSIENT = ("engine", "A", "G", "_gram_ws", "_gram", "_v", "_info", "_qp_bad", "host_qp", "stats", "opt", "dropout_masks")

    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k not in self._TRANSIENT}

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.device = torch.device(self.device)
        self.net = self.net.to(self.device)
        self._bind()
        self.opt = None

    def compute_offsets(self, task_idx, cum_nc_per_task):
        return compute_offsets(task_idx, cum_nc_per_task)

    def parameters(self):
        return self.net.parameters()

    def eval(self):
        return self

    def to(self, device):
        return self

    # ------------------------------------------------------------------ memory
    def init_new_task(self, t):
        self.observed_tasks.append(t)
        self.old_task = t

    def fill_buffer(self, t, x, y):
        """gem.py:322-345 (ring buffer; exemplar tensors instead of paths)."""
        bsz = y.shape[0]
        endcnt = min(self.mem_cnt + bsz, self.n_memories)
        eff = endcnt - self.mem_cnt
        self.memory_x[t, self.mem_cnt:endcnt] = x[:eff]
        self.memory_labels[t, self.mem_cnt:endcnt] = y[:eff]
        self.mem_cnt += eff
        if self.mem_cnt == self.n_memories:
            self.mem_cnt = 0
            return True
        return False

    def manage_memory(self, t, loader):
        """gem.py:347-368: fill the buffer from the first-task training set."""
        for x, y in loader:
            if t != self.old_task:
                self.init_new_task(t)
            if self.fill_buffer(t, x, y):
                return True
        return False

    # ------------------------------------------------------------------ kernels
    def _axpy(self, row, assign):
        check(_lib.lib().clhip_axpy(row.data_ptr(), self.A.grad.data_ptr(), self.A.numel, 1.0, int(assign), _stream()),
              "clhip_axpy")

    def gram(self, rows, to_host=True):
        m = len(rows)
        idx = (C.c_int * m)(*rows)
        check(_lib.lib().clhip_gem_gram(self.G.data_ptr(), self.G.shape[1], idx, m, self.A.numel, self._gram.data_ptr(),
                                        self._gram_ws.data_ptr(), self._gram_ws.numel(), _stream()), "clhip_gem_gram")
        return self._gram[:m * m].cpu().numpy().reshape(m, m) if to_host else self._gram

    def project_on_device(self, rows, t, eps=1e-3):
        """Violation test (gem.py:275-277), QP (:58-80) and projection / overwrite_grad (:79, :38-55) of the current
        gradient G[t] without a host round trip: Gram of rows + [t] -> clhip_gem_qp -> clhip_gem_project_dev.  Returns the
        device counter of violated constraints (0 => the gradient was left as it is)."""
        m = len(rows) + 1
        self.gram(list(rows) + [t], to_host=False)
        L = _lib.lib()
        check(L.clhip_gem_qp(self._gram.data_ptr(), m, float(self.margin), float(eps), self._v.data_ptr(), self._info.data_ptr(),
                             _stream()), "clhip_gem_qp")
        idx = (C.c_int * (m - 1))(*rows)
        check(L.clhip_gem_project_dev(self.G.data_ptr(), self.G.shape[1], idx, self._v.data_ptr(), self._info.data_ptr(), m - 1,
                                      self.G[t].data_ptr(), self.A.grad.data_ptr(), self.A.numel, _stream()),
              "clhip_gem_project_dev")
        self._qp_bad += self._info[1:2]          # status 1 (iteration limit) / 2 (infeasible) must not pass silently
        return self._info[0].clone()

    def check_qp_status(self):
        """Raise if any projection since the last call ended without a solution (the reference's quadprog raises in that
        batch; here the status is a device counter read once per epoch, no synchronisation per batch)."""
        bad = int(self._qp_bad.item())
        self._qp_bad.zero_()
        if bad:
            raise RuntimeError("GEM: project2cone2's QP reported iteration limit / infeasibility (status sum %d)" % bad)

    def project(self, rows, v, t):
        m = len(rows)
        idx = (C.c_int * m)(*rows)
        vv = (C.c_float * m)(*[float(a) for a in v])
        check(_lib.lib().clhip_gem_project(self.G.data_ptr(), self.G.shape[1], idx, vv, m, self.G[t].data_ptr(),
                                           self.A.grad.data_ptr(), self.A.numel, _stream()), "clhip_gem_project")

    # ------------------------------------------------------------------ gem.py:206-287
    def observe(self, x, t, y):
        batch_stats = {"projected_grads": [0]}
        if t != self.old_task:
            self.init_new_task(t)
        self.fill_buffer(t, x, y)
        self.reset_dropout_config()                                   # gem.py:214-215: net.train(); fresh masks per observe
        self._dropout(True)
        if len(self.observed_tasks) > 1:
            for past in self.observed_tasks[:-1]:
                sl = compute_offsets(past, self.cum_nc_per_task)
                mem = TensorTaskDataset.__new__(TensorTaskDataset)
                mem.x, mem.y, mem.classes = self.memory_x[past], self.memory_labels[past], []
                first = True
                for xb, yb in DeviceLoader(mem, self.batch_size, True, self.device):
                    self.engine.loss_step(xb.contiguous(), yb.contiguous(), "ce_mean", True, class_slice=sl)
                    self._axpy(self.G[past], assign=first)          # grads accumulate over batches (:237-256)
                    first = False
        sl = compute_offsets(t, self.cum_nc_per_task)
        self.stats.zero_()
        loss, _ = self.engine.loss_step(x, y, "ce_mean", True, self.stats, class_slice=sl)
        loss = loss.clone()
        if len(self.observed_tasks) > 1:
            self._axpy(self.G[t], assign=True)                       # store_grad (:272)
            rows = list(self.observed_tasks[:-1]) + [t]
            if self.host_qp is None:
                batch_stats["projected_grads"] = [self.project_on_device(rows[:-1], t)]     # device counter, no sync
            else:                                                     # injected host solver (cross-check in the tests)
                gram = self.gram(rows)
                dotp = gram[-1, :-1]                                  # g . G_tt (:275-276)
                viol = int((dotp < 0).sum())
                if viol != 0:
                    batch_stats["projected_grads"] = [viol]
                    v = self.host_qp(gram, len(rows) - 1, list(range(len(rows) - 1)), self.margin)
                    self.project(rows[:-1], v, t)                     # project2cone2 + overwrite_grad (:278-283)
        self.opt.step()
        return loss, self.stats[1], batch_stats

    def observe_FT(self, x, t, y):
        """gem.py:289-309: plain SGD step on the task's output slice (phase-1 grid; no memory)."""
        sl = compute_offsets(t, self.cum_nc_per_task)
        self.stats.zero_()
        self._dropout(True)       # no reset here: the masks drawn after init_setup stay for the whole run, as in the reference
        loss, _ = self.engine.loss_step(x, y, "ce_mean", True, self.stats, class_slice=sl)
        self.opt.step()
        return loss, self.stats[1]

    def eval_batch(self, x, y, t, stats):
        """main_rehearsal.py:18-35: CE and hits on the task slice (accumulated into stats on the device)."""
        sl = compute_offsets(t, self.cum_nc_per_task)
        self._dropout(False)
        return self.engine.loss_step(x, y, "ce_mean", False, stats, class_slice=sl)[0]

    def __call__(self, x, t, **kw):
        return self.forward(x, t)

    def forward(self, x, t):
        """gem.py:169-204 (eval): logits with everything outside the task slice at -1e11."""
        self._dropout(False)
        logits = self.engine.forward(x)
        o1, o2 = compute_offsets(t, self.cum_nc_per_task)
        out = torch.full_like(logits, -10e10)
        out[:, o1:o2] = logits[:, o1:o2]
        return out
