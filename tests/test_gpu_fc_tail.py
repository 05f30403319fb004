"""The fused classifier tail (fc_tail_kernel: Linear 2..3 forward, cross-entropy, backward-data down to dz(h1) in one
launch) against the per-layer launches it replaces (CLHIP_FC_TAIL=0 at plan creation): BITWISE equal logits, loss, hit
counts and gradients — the end-to-end fixtures are chaotic in the last bit, so the fused kernel keeps the k-order, the
padding, the epilogue order and the loss summation order of the GEMM / loss kernels."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _engine(model, batch, hw, tail):
    from clsurvey_amd.net import NetEngine
    old = os.environ.get("CLHIP_FC_TAIL")
    os.environ["CLHIP_FC_TAIL"] = "1" if tail else "0"
    try:
        return NetEngine(model, batch, (3, hw, hw), DEV)
    finally:
        if old is None:
            del os.environ["CLHIP_FC_TAIL"]
        else:
            os.environ["CLHIP_FC_TAIL"] = old


def _pair(name, hw, classes, batch, seed=0):
    import copy
    from clsurvey_amd import models
    torch.manual_seed(seed)
    m = models.parse_model_name(name, (hw, hw), classes)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Linear):
            torch.nn.init.kaiming_normal_(mod.weight, nonlinearity="relu")
            torch.nn.init.normal_(mod.bias, std=0.1)
    return _engine(m, batch, hw, True), _engine(copy.deepcopy(m), batch, hw, False)


@pytest.mark.parametrize("name,hw,classes,batch,n", [
    ("small_VGG9_cl_128_128", 64, 20, 200, 200),       # the headline configuration
    ("small_VGG9_cl_128_128", 64, 20, 200, 37),        # ragged last batch: 2 row blocks, the second with 5 rows
    ("small_VGG9_cl_128_128", 32, 10, 64, 64),
    ("small_VGG9_cl_128_128", 32, 32, 96, 33),         # 32 logits: the widest head the tail takes
])
def test_tail_is_bitwise_the_per_layer_path(name, hw, classes, batch, n):
    fused, plain = _pair(name, hw, classes, batch)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, 3, hw, hw, generator=g).to(DEV)
    y = torch.randint(0, classes, (n,), generator=g).to(DEV)
    for kind in ("ce_mean", "ce_sum"):
        for sl in (None, (0, classes // 2), (classes // 2, classes)):
            out = []
            for eng in (fused, plain):
                st = torch.zeros(2, dtype=torch.float64, device=DEV)
                eng.arena.grad.zero_()
                loss, logits = eng.loss_step(x, y if sl is None else y % (sl[1] - sl[0]), kind, backward=True, stats=st,
                                             want_logits=True, class_slice=sl)
                torch.cuda.synchronize()
                out.append((loss.clone(), logits.clone(), eng.arena.grad.clone(), st.clone()))
            (l0, z0, g0, s0), (l1, z1, g1, s1) = out
            assert torch.equal(z0, z1), (kind, sl)
            assert torch.equal(l0, l1) and torch.equal(s0, s1), (kind, sl, l0, l1, s0, s1)
            assert torch.equal(g0, g1), (kind, sl, float((g0 - g1).abs().max()))
            assert float(g0.abs().max()) > 0 and bool(torch.isfinite(g0).all())
    # evaluation (no backward) and plain forward take the tail too
    st0, st1 = (torch.zeros(2, dtype=torch.float64, device=DEV) for _ in range(2))
    la, _ = fused.loss_step(x, y, "ce_mean", backward=False, stats=st0)
    lb, _ = plain.loss_step(x, y, "ce_mean", backward=False, stats=st1)
    assert torch.equal(la, lb) and torch.equal(st0, st1)
    assert torch.equal(fused.forward(x), plain.forward(x))
    # repeated launches: the arrival counter resets itself
    for _ in range(3):
        lc, _ = fused.loss_step(x, y, "ce_mean", backward=True)
        assert torch.equal(lc, lb)


def test_tail_steps_aside_for_dropout_and_wide_heads():
    """Dropout inside the classifier, more than 32 logits, or 512-wide hidden layers run the per-layer launches (same
    results as before: covered by the parity tests); here only that both kinds of plan agree on such models."""
    for name, classes in (("small_VGG9_cl_128_128_DROP", 10), ("small_VGG9_cl_128_128", 40), ("base_VGG9_cl_512_512", 10)):
        fused, plain = _pair(name, 32, classes, 16)
        g = torch.Generator().manual_seed(9)
        x = torch.randn(16, 3, 32, 32, generator=g).to(DEV)
        y = torch.randint(0, classes, (16,), generator=g).to(DEV)
        for eng in (fused, plain):
            eng.model.eval()
        la, za = fused.loss_step(x, y, "ce_mean", backward=True, want_logits=True)
        lb, zb = plain.loss_step(x, y, "ce_mean", backward=True, want_logits=True)
        assert torch.equal(za, zb) and torch.equal(la, lb) and torch.equal(fused.arena.grad, plain.arena.grad)
