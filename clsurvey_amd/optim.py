"""Host-side mirror of the reference's custom optimizers, running on the fused HIP kernels.

Same class names, constructor arguments and step() signatures as
  Weight_Regularized_SGD  EWC/train_EWC.py:12-86, MAS/train_MAS.py:21-95
  Objective_After_SGD     MAS/train_MAS.py:128-181
  Elastic_SGD             SI/train_SI.py:20-126
so trainer code reads like the reference's.  When the parameters live in a ParamArena
(clsurvey_amd.net) and reg_params has been laid out with `arena_reg_params`, one step is ONE kernel
over the arena; otherwise it is one kernel per parameter tensor.  Either way there is no torch-op
fallback: CPU tensors raise.
"""
import torch
from torch.optim import Optimizer

from . import ops


def _arena_of(params):
    from .net import ParamArena
    return ParamArena.find(params)


class _SGDBase(Optimizer):
    def __init__(self, params, lr=0.001, momentum=0, dampening=0, weight_decay=0, nesterov=False):
        if dampening != 0 or nesterov:
            raise NotImplementedError("reference never uses dampening/nesterov on this path")
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov)
        super().__init__(params, defaults)
        self._steps = 0
        self._arena = None

    def _group(self):
        if len(self.param_groups) != 1:
            raise NotImplementedError("single param group only (as in the reference)")
        return self.param_groups[0]

    def arena(self):
        if self._arena is None:
            self._arena = _arena_of(self._group()["params"]) or False
        return self._arena or None

    def _buf(self, p, arena):
        st = self.state[p]
        first = "momentum_buffer" not in st
        if first:
            st["momentum_buffer"] = arena.view("buf", p) if arena is not None else torch.zeros_like(p.data)
        return st["momentum_buffer"], first

    def load_state_dict(self, state_dict):
        """Resume (train_SGD.py:62-75): torch restores `momentum_buffer` as fresh tensors; the fused arena kernels read
        the momentum from the arena's 'buf' region, so copy the restored values there, rebind the state entries to the
        arena views and mark the optimizer as past its first step (first=True would overwrite the buffer with g)."""
        super().load_state_dict(state_dict)
        restored = [p for p in self._group()["params"] if "momentum_buffer" in self.state.get(p, {})]
        if not restored:
            return
        a = self.arena()
        if a is not None:
            a.buffer("buf")                     # allocate (zeroed) if this optimizer has not stepped yet
            for p in restored:
                view = a.view("buf", p)
                view.copy_(self.state[p]["momentum_buffer"].to(view.device))
                self.state[p]["momentum_buffer"] = view
        self._steps = max(self._steps, 1)

    def zero_grad(self, set_to_none=False):
        a = self.arena()
        if a is not None:
            a.zero_grad()
        else:
            super().zero_grad(set_to_none=False)


def _reg_layout(arena, reg_params, names):
    """True if reg_params' tensors are the arena's aux views (fast path)."""
    for p in arena.params:
        rp = reg_params.get(p)
        if rp is None:
            continue
        for n in names:
            if rp[n].data_ptr() != arena.view(n, p).data_ptr():
                return False
    return True


class Weight_Regularized_SGD(_SGDBase):
    """EWC / MAS penalised momentum SGD: step(reg_params) — train_EWC.py:23-86."""

    def step(self, reg_params, closure=None):
        loss = closure() if closure is not None else None
        g = self._group()
        lam = reg_params.get("lambda")
        lr, mom, wd = g["lr"], g["momentum"], g["weight_decay"]
        a = self.arena()
        if a is not None and reg_params.get("__arena__") is a and mom != 0:
            first = self._steps == 0
            if first:
                for p in g["params"]:
                    self._buf(p, a)
            ops.reg_sgd_step(a.theta, a.grad, a.aux["omega"], a.aux["init_val"], a.buffer("buf", zero=False),
                             lam, lr, mom, wd, first)
        else:
            for p in g["params"]:
                if p.grad is None:
                    continue
                rp = reg_params.get(p)
                buf, first = self._buf(p, None) if mom != 0 else (torch.zeros_like(p.data), True)
                ops.reg_sgd_step(p.data, p.grad.data, rp["omega"] if rp is not None else None,
                                 rp["init_val"] if rp is not None else None, buf, lam, lr, mom, wd, first or mom == 0)
        self._steps += 1
        return loss


class Elastic_SGD(_SGDBase):
    """SI: penalised momentum SGD + path integral — train_SI.py:28-126. Every param must be in
    reg_params (train_SI.py:57-62 has no membership test)."""

    def step(self, reg_params, closure=None):
        loss = closure() if closure is not None else None
        g = self._group()
        lam = reg_params.get("lambda")
        lr, mom, wd = g["lr"], g["momentum"], g["weight_decay"]
        a = self.arena()
        if a is not None and reg_params.get("__arena__") is a and mom != 0:
            first = self._steps == 0
            if first:
                for p in g["params"]:
                    self._buf(p, a)
            ops.si_step(a.theta, a.grad, a.aux["omega"], a.aux["init_val"], a.aux["w"], a.buffer("buf", zero=False),
                        lam, lr, mom, wd, first)
        else:
            for p in g["params"]:
                if p.grad is None:
                    continue
                rp = reg_params[p]
                buf, first = self._buf(p, None) if mom != 0 else (torch.zeros_like(p.data), True)
                ops.si_step(p.data, p.grad.data, rp["omega"], rp["init_val"], rp["w"], buf, lam, lr, mom, wd,
                            first or mom == 0)
        self._steps += 1
        return loss


class Objective_After_SGD(_SGDBase):
    """MAS importance accumulation: step(reg_params, batch_index, batch_size) — train_MAS.py:138-181.
    Never moves the parameters."""

    def step(self, reg_params, batch_index, batch_size, closure=None):
        loss = closure() if closure is not None else None
        g = self._group()
        a = self.arena()
        if a is not None and reg_params.get("__arena__") is a:
            ops.mas_accum(a.aux["omega"], a.grad, batch_index, batch_size)
        else:
            for p in g["params"]:
                if p.grad is None or p not in reg_params:
                    continue
                ops.mas_accum(reg_params[p]["omega"], p.grad.data, batch_index, batch_size)
        self._steps += 1
        return loss


class SGD(_SGDBase):
    """optim.SGD(momentum=0.9, weight_decay) of Finetune/main_SGD.py:73 on the same fused kernel."""

    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self._group()
        lr, mom, wd = g["lr"], g["momentum"], g["weight_decay"]
        a = self.arena()
        if a is not None and mom != 0:
            first = self._steps == 0
            if first:
                for p in g["params"]:
                    self._buf(p, a)
            ops.reg_sgd_step(a.theta, a.grad, None, None, a.buffer("buf", zero=False), 0.0, lr, mom, wd, first)
        else:
            for p in g["params"]:
                if p.grad is None:
                    continue
                buf, first = self._buf(p, None) if mom != 0 else (torch.zeros_like(p.data), True)
                ops.reg_sgd_step(p.data, p.grad.data, None, None, buf, 0.0, lr, mom, wd, first or mom == 0)
        self._steps += 1
        return loss


def arena_reg_params(arena, reg_params, names=("omega", "init_val")):
    """Move the per-parameter tensors of a reference-style reg_params dict into arena aux buffers and
    replace the dict entries by views of them (object identity of keys is preserved). Parameters
    without an entry (fresh head) get zeros in the arena => identical arithmetic to 'p not in
    reg_params' (2*lambda*0*(theta-0) = 0)."""
    for n in names:
        arena.load(n, {p: reg_params[p][n] for p in arena.params if p in reg_params and n in reg_params[p]})
        for p in arena.params:
            if p in reg_params:
                reg_params[p][n] = arena.view(n, p)
    reg_params["__arena__"] = arena
    return reg_params


def detach_reg_params(reg_params):
    """Drop the arena marker before pickling (the views stay valid tensors)."""
    reg_params.pop("__arena__", None)
    return reg_params


class Adadelta(Optimizer):
    """torch.optim.Adadelta(params, lr, rho=0.9, eps=1e-6, weight_decay=0) — the optimizer of EBLL's autoencoder
    (EBLL/Finetune_SGD_EBLL.py:497) — one fused kernel per parameter tensor (an autoencoder has four)."""

    def __init__(self, params, lr=1.0, rho=0.9, eps=1e-6, weight_decay=0):
        super().__init__(params, dict(lr=lr, rho=rho, eps=eps, weight_decay=weight_decay))

    def step(self, closure=None):
        from ._lib import check, lib
        loss = closure() if closure is not None else None
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["square_avg"] = torch.zeros_like(p.data)
                    st["acc_delta"] = torch.zeros_like(p.data)
                st["step"] += 1
                if not (p.data.is_cuda and p.data.is_contiguous() and p.grad.is_contiguous()):
                    raise RuntimeError("Adadelta needs contiguous HIP tensors (no CPU fallback)")
                check(lib().clhip_adadelta_step(p.data.data_ptr(), p.grad.data_ptr(), st["square_avg"].data_ptr(),
                                                st["acc_delta"].data_ptr(), p.numel(), float(g["lr"]), float(g["rho"]),
                                                float(g["eps"]), float(g["weight_decay"]),
                                                torch.cuda.current_stream().cuda_stream), "clhip_adadelta_step")
        return loss
