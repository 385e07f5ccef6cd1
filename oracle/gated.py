"""Gate-forced fp64 evaluations of the discriminator and the VGG feature net.

TEST INFRASTRUCTURE ONLY (see oracle/sr_oracle.py header).

Why: ReLU / LeakyReLU / max-pool are piecewise linear.  A pre-activation within fp32 round-off of zero (or two
window elements within round-off of each other) may take either branch under two equally valid fp32 summation
orders, which changes a whole receptive field of a gradient by O(1).  Comparing gradients of two fp32
implementations therefore needs either loose norms (which can hide a real bug) or -- what this module does --
an fp64 evaluation that is told which branch the implementation under test took (its "gates", read from its own
saved activations) and otherwise recomputes everything independently in double precision.  With the gates pinned
the function is smooth, so every gradient must then agree to fp32 round-off (1e-5 .. 1e-4 relative, max norm).
The restated arithmetic is the same as sr_oracle.disc_vgg_forward / vgg19_conv54 (same reference citations).
"""
import torch
import torch.nn.functional as F

from . import sr_oracle as O


def _gate(pre, pos_mask, slope):
    """activation(pre) with the branch chosen by pos_mask (bool, same shape)."""
    return pre * torch.where(pos_mask, torch.ones((), dtype=pre.dtype), torch.full((), slope, dtype=pre.dtype))


def disc_vgg_forward_gated(x, sd, size, base_nf, gates, hid_gate, eps=1e-5):
    """sr_oracle.disc_vgg_forward (training-mode BatchNorm, batch statistics) in the dtype of x / sd, with every
    LeakyReLU branch taken from `gates` (one bool NCHW tensor per conv layer: activation output > 0) and
    `hid_gate` ([N, hidden] bool) for the classifier.  Running statistics are not touched."""
    convs, nc, cur = O.disc_vgg_layout(size, base_nf)
    assert len(gates) == len(convs)
    for (i, _cin, _cout, k, s, bn), g in zip(convs, gates):
        x = O._conv(x, sd, "features.%d" % i, stride=s, pad=1)
        if bn:
            p = "features.%d" % (i + 1)
            x = F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.0, eps)
        x = _gate(x, g, O.LRELU)
    x = x.reshape(x.shape[0], -1)
    x = _gate(F.linear(x, sd["classifier.0.weight"], sd["classifier.0.bias"]), hid_gate, O.LRELU)
    return F.linear(x, sd["classifier.2.weight"], sd["classifier.2.bias"])


def vgg19_conv54_gated(x, sd, relu_gates, pool_indices):
    """sr_oracle.vgg19_conv54 with ReLU branches from `relu_gates` {conv name: bool NCHW (output > 0)} and max-pool
    winners from `pool_indices` {pool name: int64 indices as returned by F.max_pool2d(..., return_indices=True)}."""
    mean = torch.tensor(O.IMAGENET_MEAN, dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor(O.IMAGENET_STD, dtype=x.dtype).view(1, 3, 1, 1)
    x = (x - mean) / std
    last = O.VGG19_NAMES[-1]
    for name in O.VGG19_NAMES:
        if name.startswith("pool"):
            idx = pool_indices[name]
            N, C, H, W = x.shape
            x = x.reshape(N, C, H * W).gather(2, idx.reshape(N, C, -1)).reshape(N, C, H // 2, W // 2)
        else:
            x = O._conv(x, sd, "feature_net." + name)
            if name != last:
                x = x * relu_gates[name].to(x.dtype)
    return x


def unet_disc_forward_gated(x, sd, gates, skip_connection=True):
    """sr_oracle.unet_disc_forward with every LeakyReLU branch taken from `gates` (9 bool NCHW tensors: the outputs of
    conv0 .. conv8 before any skip add, > 0)."""
    def c(t, i, stride=1):
        return F.conv2d(t, sd["conv%d.weight" % i], sd.get("conv%d.bias" % i), stride=stride, padding=1)

    def up(t):
        return F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False)

    x0 = _gate(c(x, 0), gates[0], O.LRELU)
    x1 = _gate(c(x0, 1, 2), gates[1], O.LRELU)
    x2 = _gate(c(x1, 2, 2), gates[2], O.LRELU)
    x3 = _gate(c(x2, 3, 2), gates[3], O.LRELU)
    x4 = _gate(c(up(x3), 4), gates[4], O.LRELU)
    if skip_connection:
        x4 = x4 + x2
    x5 = _gate(c(up(x4), 5), gates[5], O.LRELU)
    if skip_connection:
        x5 = x5 + x1
    x6 = _gate(c(up(x5), 6), gates[6], O.LRELU)
    if skip_connection:
        x6 = x6 + x0
    out = _gate(c(x6, 7), gates[7], O.LRELU)
    out = _gate(c(out, 8), gates[8], O.LRELU)
    return c(out, 9)


def gates_of_unet(saved):
    """LeakyReLU gates of trainner_amd's UNetDiscriminator from its saved activations (enc = x0..x3, ys = the decoder
    activations BEFORE the skip add, o7, o8)."""
    acts = list(saved["enc"]) + list(saved["ys"]) + [saved["o7"], saved["o8"]]
    return [a.dense().permute(0, 3, 1, 2).detach().cpu() > 0 for a in acts]


def gates_of_discriminator(saved):
    """LeakyReLU gates of trainner_amd's Discriminator_VGG from the activations its forward saved (engine `saved`
    dict: acts = [(input view, pre-BN view or None, activation view, mean, invstd)], hid = classifier hidden)."""
    gates = [a[2].dense().permute(0, 3, 1, 2).detach().cpu() > 0 for a in saved["acts"]]
    return gates, saved["hid"].detach().cpu() > 0


def gates_of_vgg(saved):
    """ReLU gates and max-pool winners of trainner_amd's FeatureExtractor from its saved tape [(name, in, out)]."""
    relu, pools = {}, {}
    for name, xin, y in saved["tape"]:
        if name.startswith("conv"):
            relu[name] = y.dense().permute(0, 3, 1, 2).detach().cpu() > 0
        else:
            xi = xin.dense().permute(0, 3, 1, 2).detach().cpu().contiguous()
            pools[name] = F.max_pool2d(xi, 2, 2, return_indices=True)[1]
    return relu, pools
