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
    # the one-grid backward of a small 3x3 layer: workspace = transformed weights + weight-gradient slabs; 0 for channel counts the
    # Winograd weight gradient does not take; argument errors before any launch
    assert lib.clhip_conv3x3_wino_bwd_ws(200, 64, 64, 16, 16) > lib.clhip_conv3x3_wino_bwd_weight_ws(200, 64, 64, 16, 16) > 0
    assert lib.clhip_conv3x3_wino_bwd_ws(200, 48, 64, 16, 16) == 0
    assert lib.clhip_conv3x3_wino_bwd(None, None, None, None, None, None, None, None, 8, 64, 64, 16, 16, None, 0, None) == -1


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


def test_parse_alexnet_plan_on_cpu():
    """torchvision-shaped AlexNet: geometry and the Dropout positions of the static plan; the oracle restatement agrees
    with the nn.Sequential modules themselves in eval mode."""
    import torch
    from clsurvey_amd import models, net
    from oracle import alexnet_ref
    m = models.AlexNet(num_classes=10, widths=(8, 12, 16, 16, 8), fc=32, feat_hw=1)
    layers, drops = net.parse_net(m)
    assert [(k, p) for k, _, _, p in layers] == [("conv", (3, 2)), ("conv", (3, 2)), ("conv", False), ("conv", False),
                                                ("conv", (3, 2)), ("fc", False), ("fc", False), ("fc", False)]
    assert [net.conv_geometry(mod) for k, mod, _, _ in layers if k == "conv"] == [(11, 4, 2), (5, 1, 2), (3, 1, 1), (3, 1, 1), (3, 1, 1)]
    assert sorted(drops) == [5, 6]
    full = models.parse_model_name("alexnet_pretrained_imgnet", num_classes=None)
    assert full.classifier[6].out_features == 1000 and full.classifier[1].in_features == 9216
    x = torch.randn(2, 3, 67, 67)
    m.eval()
    with torch.no_grad():
        want = m.classifier(torch.flatten(m.features(x), 1))
        got = alexnet_ref.forward(m, x)
    assert torch.allclose(got, want, atol=1e-6)


def test_parse_drop_variant_on_cpu():
    from clsurvey_amd import models, net
    import pytest
    m = models.parse_model_name("base_VGG9_cl_512_512_DROP", (64, 64), 20)
    layers, drops = net.parse_net(m)
    assert len(layers) == 9 and sorted(drops) == [7, 8] and len(m.classifier._modules) - 1 == 6
    bn = models.parse_model_name("base_VGG9_cl_512_512_DROP_BN", (64, 64), 20)
    with pytest.raises(NotImplementedError):
        net.parse_net(bn)                     # callers that do not handle BatchNorm say so
    layers, drops, bns = net.parse_net(bn, with_bn=True)
    assert sorted(bns) == [0, 1, 2, 3, 4, 5] and sorted(drops) == [7, 8] and all(r for _, _, r, _ in layers[:6])
    assert [k for k, *_ in layers] == ["conv"] * 6 + ["fc"] * 3
