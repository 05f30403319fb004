"""VGGSlim-compatible model builder (no torchvision needed).

Mirrors /root/reference/src/models/VGGSlim.py:19-76 and models/net.py:127-170: the module tree
is `features` (Sequential of Conv2d/ReLU/MaxPool2d), `avgpool` (Identity) and `classifier`
(Linear-ReLU-Linear-ReLU-Linear), so `model.classifier._modules[str(idx)]` indexing done by the
framework (utilities/utils.py:68-72, methods/method.py:232) works unchanged and pickles hold only
standard torch modules.  forward() runs on the HIP kernels of libclhip through autograd bridges.
"""
import torch
import torch.nn as nn

from . import ops

CFG = {  # models/VGGSlim.py:19-23
    "small_VGG9": [64, "M", 64, "M", 64, 64, "M", 128, 128, "M"],
    "base_VGG9": [64, "M", 64, "M", 128, 128, "M", 256, 256, "M"],
    "wide_VGG9": [64, "M", 128, "M", 256, 256, "M", 512, 512, "M"],
    "deep_VGG22": [64, "M", 64, 64, 64, 64, 64, 64, "M", 128, 128, 128, 128, 128, 128, "M",
                   256, 256, 256, 256, 256, 256, "M"],
}


def make_layers(cfg, in_channels=3):
    layers = []
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(in_channels, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            in_channels = v
    return nn.Sequential(*layers)


class VGGSlim(nn.Module):
    def __init__(self, config="small_VGG9", num_classes=20, init_weights=True, classifier_inputdim=128 * 4 * 4,
                 classifier_dim1=128, classifier_dim2=128, cfg=None):
        super().__init__()
        self.features = make_layers(cfg if cfg is not None else CFG[config])
        self.avgpool = nn.Identity()
        self.classifier = nn.Sequential(
            nn.Linear(classifier_inputdim, classifier_dim1), nn.ReLU(True),
            nn.Linear(classifier_dim1, classifier_dim2), nn.ReLU(True),
            nn.Linear(classifier_dim2, num_classes))
        if init_weights:
            self._initialize_weights()

    def _initialize_weights(self):
        # torchvision VGG._initialize_weights (called from VGGSlim.py:75-76)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        mods = list(self.features.children())
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Conv2d):
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                x = ops.conv3x3_relu(x, m.weight, m.bias, relu)
                i += 2 if relu else 1
            elif isinstance(m, nn.MaxPool2d):
                x = ops.maxpool2(x)
                i += 1
            else:
                raise NotImplementedError(type(m))
        x = torch.flatten(x, 1)
        mods = list(self.classifier.children())
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Linear):
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                x = ops.linear(x, m.weight, m.bias, relu)
                i += 2 if relu else 1
            else:
                raise NotImplementedError(type(m))
        return x


def parse_model_name(name, input_size=(64, 64), num_classes=20):
    """'small_VGG9_cl_128_128' style names — models/net.py:127-170."""
    base = name.split("_cl_")[0]
    dims = name.split("_cl_")[1].split("_") if "_cl_" in name else ["512", "512"]
    d1, d2 = int(dims[0]), int(dims[1])
    cfg = CFG[base]
    last = [v for v in cfg if v != "M"][-1]
    npool = sum(1 for v in cfg if v == "M")
    feat = last * (input_size[0] // 2 ** npool) * (input_size[1] // 2 ** npool)
    return VGGSlim(config=base, num_classes=num_classes, classifier_inputdim=feat,
                   classifier_dim1=d1, classifier_dim2=d2)
