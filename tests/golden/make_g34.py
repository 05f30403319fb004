"""G34: the regularised / deep model NAMES through the reference's own model factory (dev container only).

models/net.py:15-36 `parse_model_name` + :133-175 `VGGModel` + :243-262 `make_VGGmodel` + models/VGGSlim.py:27-76, UNCHANGED,
create and pickle the base model of
    small_VGG9_cl_128_128_BN, small_VGG9_cl_128_128_DROP, small_VGG9_cl_128_128_DROP_BN, deep_VGG22_cl_512_512
for 32 x 32 inputs; the fixture holds, per name,
  * the module tree as [(qualified name, class)] and the factory's `last_layer_idx` (head surgery index, net.py:139,153),
  * the initialisation the factory left: per parameter (mean, std, min, max) — torchvision's `_initialize_weights`
    (Kaiming fan-out convolutions, N(0, 0.01) Linear, BatchNorm 1 / 0, zero biases),
  * ONE TRAINING-MODE STEP of that very module object (model.train(): BatchNorm batch statistics + running-stat update,
    nn.Dropout active): parameters re-filled from a seed on both sides (g20_common.fill_params), batch from a seed, the
    masks the reference's own nn.Dropout modules drew (read back through forward hooks: out / in), logits, mean
    cross-entropy loss, every gradient (all elements of tensors up to 2^16, a fixed sample + float64 checksums of the larger
    ones), BatchNorm running statistics after the step.

    python tests/golden/make_g34.py
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "harness"))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import harness  # noqa: E402

torch = harness.install()
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import utilities.utils as utils  # noqa: E402
import g20_common as C  # noqa: E402

NAMES = ["small_VGG9_cl_128_128_BN", "small_VGG9_cl_128_128_DROP", "small_VGG9_cl_128_128_DROP_BN", "deep_VGG22_cl_512_512"]
HW, NB, NCLS = 32, 6, 20
OUT = {}


def main():
    root = tempfile.mkdtemp(prefix="g34_")
    os.makedirs(os.path.join(root, "data", "models"))
    utils.get_root_src_path = lambda: root
    import models.net as ref_net
    torch.set_num_threads(8)
    for k, name in enumerate(NAMES):
        torch.manual_seed(340 + k)
        base = ref_net.parse_model_name(os.path.join(root, "data", "models"), name, (HW, HW))
        model = torch.load(base.path)
        OUT[name + "__tree"] = np.array(["%s:%s" % (n, type(m).__name__) for n, m in model.named_modules() if n])
        OUT[name + "__last_layer_idx"] = np.array([base.last_layer_idx])
        OUT[name + "__param_names"] = np.array([n for n, _ in model.named_parameters()])
        OUT[name + "__init_stats"] = np.array([[float(p.mean()), float(p.std()) if p.numel() > 1 else 0.0, float(p.min()), float(p.max())]
                                               for p in model.parameters()])
        # one training-mode step on seeded parameters
        named = [(n, tuple(p.shape)) for n, p in model.named_parameters()]
        with torch.no_grad():
            for (n, p), q in zip(model.named_parameters(), C.fill_params(named, 3400 + k)):
                if isinstance(dict(model.named_modules())[n.rsplit(".", 1)[0]], nn.BatchNorm2d):
                    q = (1.0 + 2.0 * q) if n.endswith("weight") else q        # BatchNorm scale around 1 (fill gives 0.05 N(0,1))
                p.copy_(torch.from_numpy(q))
        x, y = (torch.from_numpy(a) for a in C.batch(3450 + k, NB, HW, NCLS))
        masks = []

        def grab(mod, inp, out):
            i = inp[0].detach()
            masks.append(torch.where(i != 0, out.detach() / torch.where(i != 0, i, torch.ones_like(i)), torch.zeros_like(i)))
        hooks = [m.register_forward_hook(grab) for m in model.modules() if isinstance(m, nn.Dropout)]
        model.train()
        logits = model(x)
        loss = F.cross_entropy(logits, y)
        grads = torch.autograd.grad(loss, list(model.parameters()))
        for h in hooks:
            h.remove()
        OUT[name + "__logits"] = logits.detach().numpy().copy()
        OUT[name + "__loss"] = np.array([float(loss)])
        for i, mk in enumerate(masks):
            OUT[name + "__dropmask%d" % i] = mk.numpy().copy()
        for j, ((n, _), g) in enumerate(zip(named, grads)):
            d = C.digest(g.numpy(), 3500 + 50 * k + j)
            OUT["%s__grad_%s__v" % (name, n)], OUT["%s__grad_%s__s" % (name, n)] = d["v"], d["s"]
        for n, b in model.named_buffers():
            OUT["%s__buf_%s" % (name, n)] = b.detach().numpy().copy()
        print(name, "loss %.6f" % float(loss), "modules", len(OUT[name + "__tree"]), "dropout masks", len(masks),
              "last_layer_idx", base.last_layer_idx)
    path = os.path.join(HERE, "G34_model_names.npz")
    np.savez_compressed(path, **OUT)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
