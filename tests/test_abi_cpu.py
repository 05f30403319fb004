"""CPU checks of the drop-in boundary: libclhip.so loads without a GPU and exports every
symbol include/clhip.h declares; the ctypes table mirrors the header; argument validation works
without touching a device."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "clhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(clhip_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from clsurvey_amd import _lib
    lib = _lib.lib()
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), "missing export: " + s
    assert sorted(_lib.SIGNATURES) == syms, "ctypes table and header drifted"
    assert lib.clhip_arch() == b"gfx950"


def test_argument_errors_do_not_need_a_device():
    from clsurvey_amd import _lib
    lib = _lib.lib()
    assert lib.clhip_conv3x3_fwd(None, None, None, None, 1, 1, 1, 8, 8, 1, None) == -1
    assert lib.clhip_fisher_accum(None, None, 10, 1.0, None) == -1
    assert lib.clhip_maxpool2_fwd(None, None, None, 1, 7, 8, None) == -1
    assert lib.clhip_conv3x3_bwd_weight_ws(200, 64, 64, 32, 32) > 0
    assert lib.clhip_fc_ws(200, 2048, 128) > 0


def test_ops_refuse_cpu_tensors():
    import torch
    from clsurvey_amd import ops
    with pytest.raises(RuntimeError):
        ops.conv3x3_fwd(torch.zeros(1, 3, 8, 8), torch.zeros(4, 3, 3, 3), torch.zeros(4))


def test_plan_shapes_on_cpu():
    import ctypes as C
    from clsurvey_amd import _lib, models, net
    m = models.parse_model_name("small_VGG9_cl_128_128")
    layers = net.parse_vgg(m)
    assert [k for k, *_ in layers] == ["conv"] * 6 + ["fc"] * 3
    descs = (_lib.LayerDesc * len(layers))()
    off = 0
    for d, (kind, mod, relu, pool) in zip(descs, layers):
        d.type = 0 if kind == "conv" else 1
        d.cin = mod.in_channels if kind == "conv" else mod.in_features
        d.cout = mod.out_channels if kind == "conv" else mod.out_features
        d.relu, d.pool = int(relu), int(pool)
        d.w_off = off
        off += mod.weight.numel()
        d.b_off = off
        off += mod.bias.numel()
    assert off == 615380
    h = C.c_void_p()
    lib = _lib.lib()
    assert lib.clhip_net_create(descs, len(layers), 200, 3, 64, 64, C.byref(h)) == 0
    assert lib.clhip_net_num_classes(h) == 20
    assert lib.clhip_net_workspace_bytes(h) > 200 * 64 * 64 * 64 * 4
    lib.clhip_net_destroy(h)
    # wrong input feature count is rejected
    descs[6].cin = 1234
    assert lib.clhip_net_create(descs, len(layers), 200, 3, 64, 64, C.byref(h)) == -1
