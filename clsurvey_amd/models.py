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


def make_layers(cfg, in_channels=3, batch_norm=False):
    layers = []
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            conv = nn.Conv2d(in_channels, v, kernel_size=3, padding=1)
            layers += [conv, nn.BatchNorm2d(v), nn.ReLU(inplace=True)] if batch_norm else [conv, nn.ReLU(inplace=True)]
            in_channels = v
    return nn.Sequential(*layers)


class VGGSlim(nn.Module):
    def __init__(self, config="small_VGG9", num_classes=20, init_weights=True, classifier_inputdim=128 * 4 * 4,
                 classifier_dim1=128, classifier_dim2=128, cfg=None, dropout=False, batch_norm=False):
        super().__init__()
        self.features = make_layers(cfg if cfg is not None else CFG[config], batch_norm=batch_norm)
        self.avgpool = nn.Identity()
        if dropout:      # the '_DROP' models (VGGSlim.py:57-66): classifier indices 0..6, last_layer_idx = 6
            self.classifier = nn.Sequential(
                nn.Linear(classifier_inputdim, classifier_dim1), nn.ReLU(True), nn.Dropout(),
                nn.Linear(classifier_dim1, classifier_dim2), nn.ReLU(True), nn.Dropout(),
                nn.Linear(classifier_dim2, num_classes))
        else:
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
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        x = _walk(x, list(self.features.children()), self.training, "VGGSlim.features")
        x = torch.flatten(x, 1)
        return _walk(x, list(self.classifier.children()), self.training, "VGGSlim.classifier")


def _walk(x, mods, training, what):
    """Run a Conv2d / ReLU / MaxPool2d / Dropout / Linear module list on the HIP kernels (autograd bridges of ops.py)."""
    i = 0
    while i < len(mods):
        m = mods[i]
        relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
        if isinstance(m, nn.Conv2d):
            ks, st, pd = m.kernel_size[0], m.stride[0], m.padding[0]
            bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d) else None
            if bn is not None:
                relu = False
            x = ops.conv3x3_relu(x, m.weight, m.bias, relu) if (ks, st, pd) == (3, 1, 1) else \
                ops.conv2d_relu(x, m.weight, m.bias, st, pd, relu)
            i += 2 if relu else 1
            if bn is not None:
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                x = ops.batchnorm_relu(x, bn, relu)
                i += 2 if relu else 1
        elif isinstance(m, nn.Linear):
            x = ops.linear(x, m.weight, m.bias, relu)
            i += 2 if relu else 1
        elif isinstance(m, nn.MaxPool2d):
            k = m.kernel_size if isinstance(m.kernel_size, int) else m.kernel_size[0]
            s = m.stride if isinstance(m.stride, int) else m.stride[0]
            x = ops.maxpool2(x) if (k, s) == (2, 2) else ops.maxpool(x, k, s)
            i += 1
        elif isinstance(m, nn.Dropout):
            if training and m.p > 0:       # mask drawn on the device; the product is a plain elementwise autograd op
                keep = 1.0 - m.p
                x = x * torch.empty_like(x).bernoulli_(keep).div_(keep)
            i += 1
        else:
            raise NotImplementedError("%s: %r" % (what, type(m)))
    return x


class AlexNet(nn.Module):
    """The module tree of torchvision.models.alexnet, which the reference instantiates for its AlexNet experiments
    (models/net.py:96-125: `models.alexnet(pretrained=...)`; `last_layer_idx = 6`): features 0..12, avgpool,
    classifier 0..6 with the same indices, so head surgery (`model.classifier._modules['6']`) and pickles are
    interchangeable.  torchvision is not needed; there is no network here, so 'pretrained' weights cannot be fetched and
    the default (torch) initialisation is what a fresh instance holds."""

    def __init__(self, num_classes=1000, dropout=0.5, widths=(64, 192, 384, 256, 256), fc=4096, feat_hw=6):
        """widths / fc / feat_hw other than torchvision's are for small test nets of the same structure."""
        super().__init__()
        c1, c2, c3, c4, c5 = widths
        self.features = nn.Sequential(
            nn.Conv2d(3, c1, kernel_size=11, stride=4, padding=2), nn.ReLU(inplace=True),
            nn.MaxPool2d(kernel_size=3, stride=2),
            nn.Conv2d(c1, c2, kernel_size=5, padding=2), nn.ReLU(inplace=True),
            nn.MaxPool2d(kernel_size=3, stride=2),
            nn.Conv2d(c2, c3, kernel_size=3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(c3, c4, kernel_size=3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(c4, c5, kernel_size=3, padding=1), nn.ReLU(inplace=True),
            nn.MaxPool2d(kernel_size=3, stride=2))
        self.feat_hw = feat_hw
        self.avgpool = nn.AdaptiveAvgPool2d((feat_hw, feat_hw))
        self.classifier = nn.Sequential(
            nn.Dropout(p=dropout), nn.Linear(c5 * feat_hw * feat_hw, fc), nn.ReLU(inplace=True),
            nn.Dropout(p=dropout), nn.Linear(fc, fc), nn.ReLU(inplace=True),
            nn.Linear(fc, num_classes))

    def forward(self, x):
        x = _walk(x, list(self.features.children()), self.training, "AlexNet.features")
        if tuple(x.shape[2:]) != (self.feat_hw, self.feat_hw):
            # 224 x 224 inputs (the reference's transforms: RandomResizedCrop(224) / CenterCrop(224)) give 6 x 6 maps, on
            # which AdaptiveAvgPool2d((6, 6)) is the identity; other sizes are outside this path
            raise NotImplementedError("AlexNet: feature maps must be 6x6 (224x224 inputs), got %s" % (tuple(x.shape[2:]),))
        x = torch.flatten(x, 1)
        return _walk(x, list(self.classifier.children()), self.training, "AlexNet.classifier")


def parse_model_name(name, input_size=(64, 64), num_classes=20):
    """'small_VGG9_cl_128_128' style names — models/net.py:127-170; 'alexnet_*' — models/net.py:23-24, 96-125."""
    if "alexnet" in name:
        return AlexNet(num_classes=1000 if num_classes is None else num_classes)
    base = name.split("_cl_")[0]
    dims = name.split("_cl_")[1].split("_") if "_cl_" in name else ["512", "512"]
    flags = name.split("_")
    d1, d2 = int(dims[0]), int(dims[1])
    cfg = CFG[base]
    last = [v for v in cfg if v != "M"][-1]
    npool = sum(1 for v in cfg if v == "M")
    feat = last * (input_size[0] // 2 ** npool) * (input_size[1] // 2 ** npool)
    return VGGSlim(config=base, num_classes=num_classes, classifier_inputdim=feat,
                   classifier_dim1=d1, classifier_dim2=d2, dropout="DROP" in flags, batch_norm="BN" in flags)
