"""Building blocks of the MI355X engine's networks.

Mirrors the *naming contract* of the reference's block.py (conv_block -> nn.Sequential whose child
'0' is the convolution: codes/models/modules/architectures/block.py:214-256, and the flattening
`sequential`, :198-211) so that state_dict keys are identical, but the modules are parameter
holders only: the arithmetic is executed by hand-written HIP kernels through trainner_amd.ops
(there is no nn.Conv2d.forward anywhere on the path).
"""
import math

import torch
import torch.nn as nn

from .... import ops

_ACTS = {None: (ops.ACT_NONE, 0.0), "relu": (ops.ACT_RELU, 0.0), "leakyrelu": (ops.ACT_LRELU, 0.2),
         "lrelu": (ops.ACT_LRELU, 0.2)}


def act_code(act_type):
    """activation name -> (kernel enum, negative slope); block.act (block.py:82-102) defaults."""
    key = act_type.lower() if isinstance(act_type, str) else act_type
    if key not in _ACTS:
        raise NotImplementedError("activation layer [%s] is not implemented by the HIP engine" % act_type)
    return _ACTS[key]


class Conv2dHIP(nn.Module):
    """Weight/bias holder for one convolution (OIHW, exactly nn.Conv2d's tensors and default init).
    The class name contains 'Conv' on purpose: networks.init_weights selects layers by class name
    (codes/models/networks.py:41-54)."""

    def __init__(self, in_nc, out_nc, kernel_size=3, stride=1, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_nc, out_nc
        self.kernel_size, self.stride = kernel_size, stride
        self.weight = nn.Parameter(torch.empty(out_nc, in_nc, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_nc)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1.0 / math.sqrt(in_nc * kernel_size * kernel_size)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        raise RuntimeError("Conv2dHIP is executed by its owning network's HIP engine, not called directly")

    def extra_repr(self):
        return "%d, %d, k=%d, s=%d" % (self.in_channels, self.out_channels, self.kernel_size, self.stride)


class Marker(nn.Module):
    """Parameter-free place holder that keeps nn.Sequential indices equal to the reference's
    (activation, Upsample, PixelShuffle positions)."""

    def __init__(self, what):
        super().__init__()
        self.what = what

    def forward(self, x):
        raise RuntimeError("marker module")

    def extra_repr(self):
        return self.what


class LinearHIP(nn.Module):
    """nn.Linear's tensors (name contains 'Linear' for init_weights)."""

    def __init__(self, in_f, out_f):
        super().__init__()
        self.in_features, self.out_features = in_f, out_f
        self.weight = nn.Parameter(torch.empty(out_f, in_f))
        self.bias = nn.Parameter(torch.empty(out_f))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(in_f)
        nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        raise RuntimeError("LinearHIP is executed by its owning network's HIP engine")


class BatchNorm2dHIP(nn.BatchNorm2d):
    """nn.BatchNorm2d's parameters and buffers (affine, running stats); executed by tnr_bn_train_*."""

    def forward(self, x):
        raise RuntimeError("BatchNorm2dHIP is executed by its owning network's HIP engine")


def flat_sequential(*mods):
    """Reference `sequential` semantics: drop None, splice nested nn.Sequential children."""
    out = []
    for m in mods:
        if m is None:
            continue
        if isinstance(m, nn.Sequential):
            out.extend(m.children())
        else:
            out.append(m)
    return nn.Sequential(*out)


def conv_block(in_nc, out_nc, kernel_size=3, stride=1, norm_type=None, act_type="relu"):
    """[conv, (BatchNorm), (activation marker)] with the reference's child ordering (mode 'CNA')."""
    if norm_type not in (None, "batch"):
        raise NotImplementedError("norm_type [%s] is not implemented by the HIP engine" % norm_type)
    mods = [Conv2dHIP(in_nc, out_nc, kernel_size, stride)]
    if norm_type == "batch":
        mods.append(BatchNorm2dHIP(out_nc, affine=True))
    if act_type:
        act_code(act_type)
        mods.append(Marker("act:" + act_type))
    return nn.Sequential(*mods)


class ShortcutBlock(nn.Module):
    """x + sub(x) (block.py:184-195); `sub` holds the trunk."""

    def __init__(self, submodule):
        super().__init__()
        self.sub = submodule
