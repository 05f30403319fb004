"""PackNet on the HIP path — mirror of src/methods/packnet/{prune,packnetSGD}.py and the batch
logic of packnet/main.py:Manager.

SparsePruner keeps the reference's interface (pruning_mask / prune / make_grads_zero /
make_pruned_zero / apply_mask / make_finetuning_mask / get_biases / restore_biases) on uint8 device
masks; the k-th-magnitude cutoff is found on the device (the reference copies the layer to the CPU for
torch.kthvalue, prune.py:39).  Masks are bit-exact with the reference (tests/golden/G7).
"""
import torch
import torch.nn as nn
from torch.optim import Optimizer

from .. import _lib
from .._lib import check


def _stream():
    return torch.cuda.current_stream().cuda_stream


_kth_ws = {}


def kth_abs(weights, mask, cur, k):
    """k-th smallest |w| over {mask == cur} as a 1-element device tensor (exact)."""
    L = _lib.lib()
    dev = weights.device
    ws = _kth_ws.get(str(dev))
    if ws is None:
        ws = _kth_ws[str(dev)] = torch.zeros(L.clhip_packnet_kth_ws(), dtype=torch.uint8, device=dev)
    out = torch.zeros(1, dtype=torch.float32, device=dev)
    check(L.clhip_packnet_kth_abs(weights.data_ptr(), mask.data_ptr(), weights.numel(), int(cur), int(k),
                                  out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "clhip_packnet_kth_abs")
    return out


class SparsePruner(object):
    def __init__(self, model, prune_perc, previous_masks, train_bias, train_bn, current_dataset_idx):
        self.model = model
        self.prune_perc = prune_perc
        self.train_bias = train_bias
        self.train_bn = train_bn
        self.current_masks = None
        self.previous_masks = previous_masks
        self.current_dataset_idx = int(current_dataset_idx)

    def _layers(self):
        for module_idx, module in enumerate(self.model.shared.modules()):
            if isinstance(module, (nn.Conv2d, nn.Linear)):
                yield module_idx, module

    def pruning_mask(self, weights, previous_mask, layer_idx):
        """prune.py:24-52. Returns the new mask (in place on previous_mask, like the reference)."""
        cur = self.current_dataset_idx
        numel = int((previous_mask == cur).sum().item())
        cutoff_rank = round(self.prune_perc * numel)                   # prune.py:32 (banker's rounding)
        if cutoff_rank < 1:
            raise RuntimeError("kthvalue(): selected index k out of range")   # what torch raises in the reference
        cutoff = kth_abs(weights, previous_mask, cur, cutoff_rank)
        return previous_mask, cutoff

    def prune(self):
        """prune.py:54-71."""
        assert not self.current_masks, "Current mask is not empty? Pruning twice?"
        self.current_masks = {}
        L = _lib.lib()
        for module_idx, module in self._layers():
            w = module.weight.data
            mask, cutoff = self.pruning_mask(w, self.previous_masks[module_idx], module_idx)
            check(L.clhip_packnet_prune(w.data_ptr(), mask.data_ptr(), w.numel(), self.current_dataset_idx,
                                        cutoff.data_ptr(), _stream()), "clhip_packnet_prune")
            self.current_masks[module_idx] = mask

    def make_grads_zero(self, cuda=False):
        """prune.py:73-97."""
        assert self.current_masks
        L = _lib.lib()
        for module_idx, module in enumerate(self.model.shared.modules()):
            if isinstance(module, (nn.Conv2d, nn.Linear)):
                if module.weight.grad is not None:
                    g = module.weight.grad.data
                    check(L.clhip_mask_grad_zero(g.data_ptr(), self.current_masks[module_idx].data_ptr(), g.numel(),
                                                 self.current_dataset_idx, _stream()), "clhip_mask_grad_zero")
                    if not self.train_bias and module.bias is not None:
                        module.bias.grad.data.fill_(0)
            elif "BatchNorm" in str(type(module)) and not self.train_bn:
                module.weight.grad.data.fill_(0)
                module.bias.grad.data.fill_(0)

    def _weight_zero(self, masks, mode, idx):
        L = _lib.lib()
        for module_idx, module in self._layers():
            w = module.weight.data
            check(L.clhip_mask_weight_zero(w.data_ptr(), masks[module_idx].data_ptr(), w.numel(), mode, int(idx),
                                           _stream()), "clhip_mask_weight_zero")

    def make_pruned_zero(self):
        """prune.py:99-106."""
        assert self.current_masks
        self._weight_zero(self.current_masks, 0, 0)

    def apply_mask(self, dataset_idx, debug=False):
        """prune.py:108-118."""
        self._weight_zero(self.previous_masks, 1, dataset_idx)

    def restore_biases(self, biases):
        for module_idx, module in self._layers():
            if module.bias is not None:
                module.bias.data.copy_(biases[module_idx])

    def get_biases(self):
        return {i: m.bias.data.clone() for i, m in self._layers() if m.bias is not None}

    def make_finetuning_mask(self):
        """prune.py:141-155."""
        assert self.previous_masks
        L = _lib.lib()
        for module_idx, module in self._layers():
            mask = self.previous_masks[module_idx]
            check(L.clhip_packnet_finetune_mask(mask.data_ptr(), mask.numel(), self.current_dataset_idx, _stream()),
                  "clhip_packnet_finetune_mask")
        self.current_masks = self.previous_masks


class PacknetSGD(Optimizer):
    """packnetSGD.py:5-58: momentum SGD whose weight decay is masked by grad != 0."""

    def __init__(self, params, lr, momentum=0, dampening=0, weight_decay=0, nesterov=False, custom_L2=None):
        if dampening != 0 or nesterov:
            raise NotImplementedError
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                                      nesterov=nesterov))

    def step(self, closure=None):
        loss = closure() if closure is not None else None
        L = _lib.lib()
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                first = "momentum_buffer" not in st
                if first:
                    st["momentum_buffer"] = torch.zeros_like(p.data)
                check(L.clhip_packnet_sgd_step(p.data.data_ptr(), p.grad.data.data_ptr(),
                                               st["momentum_buffer"].data_ptr(), None, p.numel(), 0,
                                               float(group["lr"]), float(group["momentum"]),
                                               float(group["weight_decay"]), int(first or group["momentum"] == 0),
                                               _stream()), "clhip_packnet_sgd_step")
        return loss


def fused_batch_tail(theta, grad, buf, mask_u8, cur, lr, momentum, wd, first):
    """do_batch tail of packnet/main.py:187-193 in ONE kernel over a ParamArena: foreign grads -> 0,
    PacknetSGD.step, pruned weights -> 0.  mask_u8 is a uint8 arena (biases carry 255 when
    train_biases is False so their grads are zeroed too)."""
    check(_lib.lib().clhip_packnet_sgd_step(theta.data_ptr(), grad.data_ptr(), buf.data_ptr(), mask_u8.data_ptr(),
                                            theta.numel(), int(cur), float(lr), float(momentum), float(wd), int(first),
                                            _stream()), "clhip_packnet_sgd_step")
