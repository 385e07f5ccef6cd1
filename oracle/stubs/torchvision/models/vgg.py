"""Standard VGG cfg-D/E feature stacks with SEEDED weights.

ImageNet weights cannot be downloaded here, so the oracle and the HIP engine both
use He-normal weights from torch.Generator().manual_seed(1234).
The layer ordering is the public torchvision cfg 'E' (vgg19) / 'D' (vgg16):
conv3x3(pad 1)+ReLU(inplace) pairs and MaxPool2d(2, 2).
"""
import math
import torch
import torch.nn as nn

CFG = {
    "vgg16": [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"],
    "vgg19": [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M",
              512, 512, 512, 512, "M"],
}
VGG_SEED = 1234


class VGG(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        layers, c = [], 3
        for v in cfg:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(c, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                c = v
        self.features = nn.Sequential(*layers)


def seeded_fill_(features, seed=VGG_SEED):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in features:
            if isinstance(m, nn.Conv2d):
                fan_in = m.in_channels * 9
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * math.sqrt(2.0 / fan_in))
                m.bias.zero_()


def _make(name):
    net = VGG(CFG[name])
    seeded_fill_(net.features)
    return net


def vgg19(pretrained=False, **kw):
    return _make("vgg19")


def vgg16(pretrained=False, **kw):
    return _make("vgg16")
