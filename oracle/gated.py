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


# --------------------------------------------------------------------------------------
# The whole step with pinned gates (tests/test_gpu_step.py::test_step_gate_pinned_fp64_trajectory)
# --------------------------------------------------------------------------------------
def gates_of_rrdbnet(saved, nf=64, gc=32):
    """LeakyReLU gates of trainner_amd's RRDBNet (upconv) from its saved activations: every dense block is one [N,H,W,nf+4gc] NHWC
    buffer whose channel groups 1..4 hold x1..x4 AFTER the activation; the up-sampling stages and HR_conv0 keep theirs."""
    g = {"rdb": [], "up": [], "hr0": None}
    for buf in saved["bufs"]:
        t = buf.detach().cpu()
        g["rdb"].append([t[..., nf + gc * k: nf + gc * (k + 1)].permute(0, 3, 1, 2) > 0 for k in range(4)])
    for _src, dst, _t in saved["stages"]:
        g["up"].append(dst.dense().permute(0, 3, 1, 2).detach().cpu() > 0)
    g["hr0"] = saved["h0"].dense().permute(0, 3, 1, 2).detach().cpu() > 0
    return g


def rrdbnet_forward_gated(lr, sd, nb, gates, upscale=4):
    """sr_oracle.rrdbnet_forward (upconv) in the dtype of lr / sd with every LeakyReLU branch taken from `gates`;
    gates.get("noise"): the ESRGAN+ multiplier fields of the dense blocks (sr_oracle.rdb5c_forward's `m`), in that dtype."""
    import math
    n_up = int(math.log(upscale, 2))
    fea = O._conv(lr, sd, "model.0")
    t = fea
    i = 0
    for b in range(nb):
        x_rrdb = t
        for r in (1, 2, 3):
            pre, x0 = "model.1.sub.%d.RDB%d" % (b, r), t
            feats = [x0]
            for k in range(1, 5):
                feats.append(_gate(O._conv(torch.cat(feats, 1), sd, "%s.conv%d.0" % (pre, k)), gates["rdb"][i][k - 1], O.LRELU))
            t = O._conv(torch.cat(feats, 1), sd, pre + ".conv5.0") * 0.2 + x0
            if gates.get("noise") is not None:
                t = t * gates["noise"][i]
            i += 1
        t = t * 0.2 + x_rrdb
    y = fea + O._conv(t, sd, "model.1.sub.%d" % nb)
    idx = 2
    for u in range(n_up):
        y = F.interpolate(y, scale_factor=2.0, mode="nearest")
        y = _gate(O._conv(y, sd, "model.%d" % (idx + 1)), gates["up"][u], O.LRELU)
        idx += 3
    y = _gate(O._conv(y, sd, "model.%d" % idx), gates["hr0"], O.LRELU)
    return O._conv(y, sd, "model.%d" % (idx + 2))


def _bn_batch_stats(x, sd, size, base_nf, gates, eps=1e-5):
    """Per BatchNorm layer of Discriminator_VGG: (batch mean, UNBIASED batch variance) of its input in one gated forward."""
    convs, _nc, _cur = O.disc_vgg_layout(size, base_nf)
    stats = {}
    for (i, _cin, _cout, k, s, bn), g in zip(convs, gates):
        x = O._conv(x, sd, "features.%d" % i, stride=s, pad=1)
        if bn:
            p = "features.%d" % (i + 1)
            n = x.numel() // x.shape[1]
            stats[p] = (x.mean((0, 2, 3)).detach(), (x.var((0, 2, 3), unbiased=False) * n / (n - 1)).detach())
            x = F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.0, eps)
        x = _gate(x, g, O.LRELU)
    return stats


def adam64(p, g, m, v, t, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    """One torch.optim.Adam update (optimizers.py:130-132 defaults) in float64; returns (p, m, v)."""
    import math
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    denom = v.sqrt() / math.sqrt(1 - b2 ** t) + eps
    return p - (lr / (1 - b1 ** t)) * m / denom, m, v


def gated_step64(LR, HR, g_sd, d_sd, f_sd, gates, adam_state, *, nb, d_size, d_nf, pixel_weight=1e-2, feature_weight=1.0,
                 gan_weight=5e-3, grad_clip=0.1, lr=1e-4):
    """SRModel.optimize_parameters (models/sr_model.py:195-267; the ESRGAN recipe as in sr_oracle.OracleSRStep) evaluated in float64
    from the GIVEN state -- parameters g_sd / d_sd / f_sd, Adam moments and step counts in adam_state = {"G": (m, v, t), "D": ...} --
    with every piecewise-linear branch (LeakyReLU of G and D, ReLU and max-pool winners of VGG on the generated image) taken from
    `gates` = {"G": .., "D_fake": (gates, hid), "D_real": (gates, hid), "F_fake": (relu, pools)}, the branches the implementation
    under test took.  Returns logs, fake_H, the clipped G gradients, the D gradients, the updated parameters and moments, and the
    BatchNorm batch statistics of the two discriminator inputs."""
    dd = torch.float64
    G = {k: v.detach().to(dd).clone().requires_grad_(True) for k, v in g_sd.items()}
    D = {k: (v.detach().to(dd).clone().requires_grad_(O._is_param(k)) if v.is_floating_point() else v.clone()) for k, v in d_sd.items()}
    Fs = {k: v.detach().to(dd) for k, v in f_sd.items()}
    LR, HR = LR.to(dd), HR.to(dd)
    log = {}
    fake = rrdbnet_forward_gated(LR, G, nb, gates["G"])
    l_pix = pixel_weight * F.l1_loss(fake, HR)
    fx = vgg19_conv54_gated(fake, Fs, *gates["F_fake"])
    fy = O.vgg19_conv54(HR, Fs)                                    # no gradient flows through the real branch
    l_fea = F.l1_loss(fx, fy.detach()) * feature_weight
    dfix = {k: (v.detach() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in D.items()}     # D frozen in the G stage
    pf = disc_vgg_forward_gated(fake, dfix, d_size, d_nf, *gates["D_fake"])
    pr = disc_vgg_forward_gated(HR, dfix, d_size, d_nf, *gates["D_real"])
    l_gan = gan_weight * O.ragan_g_loss(pf, pr)
    log["pix-l1"], log["fea-vgg19-l1"], log["l_g_gan"] = l_pix.item(), l_fea.item(), l_gan.item()
    gk = list(G.keys())
    gg = list(torch.autograd.grad(l_pix + l_fea + l_gan, [G[k] for k in gk]))
    total = torch.sqrt(sum((g ** 2).sum() for g in gg))
    coef = torch.clamp(grad_clip / (total + 1e-6), max=1.0)
    gg = [g * coef for g in gg]
    mG, vG, tG = adam_state["G"]
    newG, newmG, newvG = {}, {}, {}
    for k, g in zip(gk, gg):
        newG[k], newmG[k], newvG[k] = adam64(G[k].detach(), g, mG[k].to(dd), vG[k].to(dd), tG + 1, lr)
    # ---- D stage on the detached fake, D parameters as before the step (the G update does not touch them)
    dk = [k for k in D if O._is_param(k)]
    pf = disc_vgg_forward_gated(fake.detach(), D, d_size, d_nf, *gates["D_fake"])
    pr = disc_vgg_forward_gated(HR, D, d_size, d_nf, *gates["D_real"])
    l_real, l_fake = O.ragan_d_loss(pf, pr)
    log["l_d_real"], log["l_d_fake"] = l_real.item(), l_fake.item()
    log["D_real"], log["D_fake"] = pr.detach().mean().item(), pf.detach().mean().item()
    dg = list(torch.autograd.grad((l_fake + l_real) * 0.5, [D[k] for k in dk]))
    mD, vD, tD = adam_state["D"]
    newD, newmD, newvD = {}, {}, {}
    for k, g in zip(dk, dg):
        newD[k], newmD[k], newvD[k] = adam64(D[k].detach(), g, mD[k].to(dd), vD[k].to(dd), tD + 1, lr)
    stats = {"fake": _bn_batch_stats(fake.detach(), dfix, d_size, d_nf, gates["D_fake"][0]),
             "real": _bn_batch_stats(HR, dfix, d_size, d_nf, gates["D_real"][0])}
    return dict(log=log, fake=fake.detach(), g_grads=dict(zip(gk, gg)), d_grads=dict(zip(dk, dg)), G=newG, D=newD,
                mG=newmG, vG=newvG, mD=newmD, vD=newvD, bn=stats)
