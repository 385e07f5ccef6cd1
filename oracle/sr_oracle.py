"""CPU restatement of the reference SR training step (the ORACLE).

TEST INFRASTRUCTURE ONLY.  Nothing under trainner_amd/ may import this module; it is used by
tests/, by `__graft_entry__.smoke()` as the checker and by bench.py's `cpu_baseline` leg
(kind "port").  It is plain fp32 PyTorch on the CPU, written functionally over state_dicts
that carry the reference's own keys, and it is PINNED against the real reference run in the
build container: tests/test_oracle_golden.py compares it with tests/golden/*.pt, which
oracle/make_golden.py produced from /root/reference itself.

Every function cites the reference lines (relative to /root/reference/codes) it restates.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

LRELU = 0.2


# --------------------------------------------------------------------------------------
# Generators
# --------------------------------------------------------------------------------------
def _conv(x, sd, key, stride=1, pad=1):
    return F.conv2d(x, sd[key + ".weight"], sd[key + ".bias"], stride=stride, padding=pad)


def rdb5c_forward(x, sd, pre, m=None):
    """ResidualDenseBlock_5C.forward (models/modules/architectures/RRDBNet_arch.py:150-163):
    four conv3x3+LeakyReLU(0.2) over growing concatenations, a fifth plain conv, x5*0.2 + x.
    m: the ESRGAN+ GaussianNoise of the block (`self.noise(x5.mul(0.2) + x)`, :160-161; block.py:594-599: x + n * (sigma * x) in
    training mode, gradient through both terms) as the multiplier field m = 1 + sigma * n -- the draw itself is an INPUT here
    (the reference takes it from torch's global generator, the engine from its counter-based field), None = no noise."""
    feats = [x]
    for i in range(1, 5):
        y = _conv(torch.cat(feats, 1), sd, "%s.conv%d.0" % (pre, i))
        feats.append(F.leaky_relu(y, LRELU))
    x5 = _conv(torch.cat(feats, 1), sd, pre + ".conv5.0")
    y = x5 * 0.2 + x
    return y if m is None else y * m


def rrdb_forward(x, sd, pre, ms=None):
    """RRDB.forward (RRDBNet_arch.py:89-96): three dense blocks then out*0.2 + x.  ms: their three noise multipliers."""
    out = x
    for r in (1, 2, 3):
        out = rdb5c_forward(out, sd, "%s.RDB%d" % (pre, r), None if ms is None else ms[r - 1])
    return out * 0.2 + x


def rrdbnet_forward(lr, sd, nb, upscale=4, upsample_mode="upconv", noise=None):
    """RRDBNet (RRDBNet_arch.py:23-49) with the flattened `B.sequential` indices
    (block.py:198-211): model.0 fea_conv; model.1 ShortcutBlock(sub.0..nb-1 RRDBs, sub.nb
    LR_conv); then per upscale stage [Upsample, conv, LeakyReLU] (upconv_block, block.py:390-404)
    or [conv, PixelShuffle, LeakyReLU] (pixelshuffle_block, block.py:374-387); HR_conv0+LeakyReLU;
    HR_conv1.  noise: 3 * nb multiplier fields [N, nf, h, w] in block order (rdb5c_forward), None = gaussian_noise off."""
    n_up = int(math.log(upscale, 2))
    fea = _conv(lr, sd, "model.0")
    t = fea
    for b in range(nb):
        t = rrdb_forward(t, sd, "model.1.sub.%d" % b, None if noise is None else noise[3 * b:3 * b + 3])
    t = _conv(t, sd, "model.1.sub.%d" % nb)
    y = fea + t                                             # ShortcutBlock (block.py:184-195)
    idx = 2
    for _ in range(n_up):
        if upsample_mode == "upconv":
            y = F.interpolate(y, scale_factor=2.0, mode="nearest")
            y = F.leaky_relu(_conv(y, sd, "model.%d" % (idx + 1)), LRELU)
        else:
            y = F.leaky_relu(F.pixel_shuffle(_conv(y, sd, "model.%d" % idx), 2), LRELU)
        idx += 3
    y = F.leaky_relu(_conv(y, sd, "model.%d" % idx), LRELU)
    return _conv(y, sd, "model.%d" % (idx + 2))


def srresnet_forward(lr, sd, nb, upscale=4):
    """SRResNet (SRResNet_arch.py:16-60) as built by defaults.py:98-112: mode CNA, no norm,
    ReLU, pixelshuffle.  ResNetBlock = conv-ReLU-conv, x + res*res_scale(1) (:62-92)."""
    n_up = int(math.log(upscale, 2))
    fea = _conv(lr, sd, "model.0")
    t = fea
    for b in range(nb):
        r = F.relu(_conv(t, sd, "model.1.sub.%d.res.0" % b))
        r = _conv(r, sd, "model.1.sub.%d.res.2" % b)
        t = t + r
    t = _conv(t, sd, "model.1.sub.%d" % nb)
    y = fea + t
    idx = 2
    for _ in range(n_up):
        y = F.relu(F.pixel_shuffle(_conv(y, sd, "model.%d" % idx), 2))
        idx += 3
    y = F.relu(_conv(y, sd, "model.%d" % idx))
    return _conv(y, sd, "model.%d" % (idx + 2))


# --------------------------------------------------------------------------------------
# Discriminator_VGG  (models/modules/architectures/discriminators.py:16-51)
# --------------------------------------------------------------------------------------
def disc_vgg_layout(size, base_nf):
    """[(features index, cin, cout, k, stride, has_bn)], final nc, final spatial size."""
    convs = [(0, None, base_nf, 3, 1, False), (2, base_nf, base_nf, 4, 2, True)]
    idx, cur, nc = 5, size // 2, base_nf
    while cur > 4:
        out = nc * 2 if nc < 512 else nc
        convs.append((idx, nc, out, 3, 1, True))
        convs.append((idx + 3, out, out, 4, 2, True))
        idx += 6
        nc, cur = out, cur // 2
    return convs, nc, cur


def disc_vgg_forward(x, sd, size, base_nf, training=True, momentum=0.1, eps=1e-5):
    """conv(+bias) -> BatchNorm2d(train: batch stats, running stats updated in place,
    torch.nn.BatchNorm2d defaults momentum 0.1 eps 1e-5) -> LeakyReLU(0.2); flatten (NCHW
    order) -> Linear(.,100) -> LeakyReLU -> Linear(100,1).  `sd` buffers are mutated the way
    nn.BatchNorm2d mutates them."""
    convs, nc, cur = disc_vgg_layout(size, base_nf)
    for (i, _cin, _cout, k, s, bn) in convs:
        x = _conv(x, sd, "features.%d" % i, stride=s, pad=1)
        if bn:
            p = "features.%d" % (i + 1)
            x = F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                             sd[p + ".bias"], training, momentum, eps)
            if training:
                sd[p + ".num_batches_tracked"] += 1
        x = F.leaky_relu(x, LRELU)
    x = x.reshape(x.shape[0], -1)
    x = F.leaky_relu(F.linear(x, sd["classifier.0.weight"], sd["classifier.0.bias"]), LRELU)
    return F.linear(x, sd["classifier.2.weight"], sd["classifier.2.bias"])


def unet_disc_forward(x, sd, skip_connection=True):
    """UNetDiscriminator.forward (models/modules/architectures/discriminators.py:736-779): conv0 (bias) + LReLU; conv1..3
    k4 s2 without bias + LReLU; three times bilinear x2 (align_corners=False) -> conv k3 without bias + LReLU -> + skip;
    conv7, conv8 + LReLU; conv9 (bias).  Keys conv<i>.weight (+ conv0.bias, conv9.bias).  Per-pixel logits [N,1,H,W]."""
    def c(t, i, stride=1):
        return F.conv2d(t, sd["conv%d.weight" % i], sd.get("conv%d.bias" % i), stride=stride, padding=1)

    def up(t):
        return F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False)

    x0 = F.leaky_relu(c(x, 0), LRELU)
    x1 = F.leaky_relu(c(x0, 1, 2), LRELU)
    x2 = F.leaky_relu(c(x1, 2, 2), LRELU)
    x3 = F.leaky_relu(c(x2, 3, 2), LRELU)
    x4 = F.leaky_relu(c(up(x3), 4), LRELU)
    if skip_connection:
        x4 = x4 + x2
    x5 = F.leaky_relu(c(up(x4), 5), LRELU)
    if skip_connection:
        x5 = x5 + x1
    x6 = F.leaky_relu(c(up(x5), 6), LRELU)
    if skip_connection:
        x6 = x6 + x0
    out = F.leaky_relu(c(x6, 7), LRELU)
    out = F.leaky_relu(c(out, 8), LRELU)
    return c(out, 9)


def _norm2d(x, sd, key, norm, training=True):
    """BatchNorm2d (train mode: batch statistics, running statistics updated in place) or InstanceNorm2d (no affine)."""
    if norm == "instance":
        return F.instance_norm(x, eps=1e-5)
    out = F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"], sd[key + ".bias"], training, 0.1, 1e-5)
    if training:
        sd[key + ".num_batches_tracked"] += 1
    return out


def resnet_generator_forward(x, sd, n_blocks, norm="instance"):
    """ResnetGenerator.forward (models/modules/architectures/ResNet_arch.py:11-149): reflect-pad 7x7 conv + norm + ReLU; two
    3x3 s2 convs + norm + ReLU; n_blocks residual blocks (reflect-pad 3x3 conv + norm + ReLU + reflect-pad 3x3 conv + norm, + x);
    two ConvTranspose2d(k3, s2, p1, op1) + norm + ReLU; reflect-pad 7x7 conv + tanh.  Keys `model.<i>.*` as in the reference."""
    def conv(t, i, stride=1, pad=0):
        return F.conv2d(t, sd["model.%d.weight" % i], sd.get("model.%d.bias" % i), stride=stride, padding=pad)

    h = F.relu(_norm2d(conv(F.pad(x, (3,) * 4, mode="reflect"), 1), sd, "model.2", norm))
    h = F.relu(_norm2d(conv(h, 4, 2, 1), sd, "model.5", norm))
    h = F.relu(_norm2d(conv(h, 7, 2, 1), sd, "model.8", norm))
    for b in range(n_blocks):
        p = "model.%d.conv_block" % (10 + b)
        t = F.conv2d(F.pad(h, (1,) * 4, mode="reflect"), sd[p + ".1.weight"], sd.get(p + ".1.bias"))
        t = F.relu(_norm2d(t, sd, p + ".2", norm))
        t = F.conv2d(F.pad(t, (1,) * 4, mode="reflect"), sd[p + ".5.weight"], sd.get(p + ".5.bias"))
        h = h + _norm2d(t, sd, p + ".6", norm)
    i = 10 + n_blocks
    for j in (i, i + 3):
        h = F.conv_transpose2d(h, sd["model.%d.weight" % j], sd.get("model.%d.bias" % j), stride=2, padding=1, output_padding=1)
        h = F.relu(_norm2d(h, sd, "model.%d" % (j + 1), norm))
    return torch.tanh(conv(F.pad(h, (3,) * 4, mode="reflect"), i + 7))


def unet_generator_forward(x, sd, num_downs, norm="batch"):
    """UnetGenerator.forward (models/modules/architectures/UNet_arch.py:11-162, deconv up-sampling, no dropout): nested blocks
    x -> cat[x, up(sub(down(x)))] with down = LeakyReLU(0.2) -> Conv2d(k4, s2, p1) -> norm and up = ReLU -> ConvTranspose2d(k4, s2,
    p1) -> norm (outermost: no activation / norm going down, bias + Tanh coming up; innermost: no norm going down).  Both
    activations are in place (:106,108), so the x a block concatenates is the ACTIVATED one.  Keys as in the reference
    (`model.model.<i>...`)."""
    def block(t, pre, level):
        outer, inner = level == 0, level == num_downs - 1
        if outer:
            h = F.conv2d(t, sd[pre + ".0.weight"], sd.get(pre + ".0.bias"), stride=2, padding=1)
            h = block(h, pre + ".1.model", level + 1)
            h = F.relu(h)
            return torch.tanh(F.conv_transpose2d(h, sd[pre + ".3.weight"], sd.get(pre + ".3.bias"), stride=2, padding=1))
        a = F.leaky_relu(t, LRELU)                          # in place in the reference: this is also what gets concatenated
        h = F.conv2d(a, sd[pre + ".1.weight"], sd.get(pre + ".1.bias"), stride=2, padding=1)
        if inner:
            h = F.relu(h)
            h = F.conv_transpose2d(h, sd[pre + ".3.weight"], sd.get(pre + ".3.bias"), stride=2, padding=1)
            h = _norm2d(h, sd, pre + ".4", norm)
        else:
            h = _norm2d(h, sd, pre + ".2", norm)
            h = block(h, pre + ".3.model", level + 1)
            h = F.relu(h)
            h = F.conv_transpose2d(h, sd[pre + ".5.weight"], sd.get(pre + ".5.bias"), stride=2, padding=1)
            h = _norm2d(h, sd, pre + ".6", norm)
        return torch.cat([a, h], 1)

    return block(x, "model.model", 0)


def patchgan_forward(x, sd, n_layers=3):
    """NLayerDiscriminator.forward (discriminators.py:472-579), default configuration: conv4 s2 + LReLU; (n_layers - 1) x
    [conv4 s2 + BatchNorm + LReLU]; conv4 s1 + BatchNorm + LReLU; conv4 s1 -> 1 channel."""
    h = F.leaky_relu(F.conv2d(x, sd["model.0.weight"], sd["model.0.bias"], stride=2, padding=1), LRELU)
    i = 2
    for n in range(1, n_layers + 1):
        stride = 2 if n < n_layers else 1
        h = F.conv2d(h, sd["model.%d.weight" % i], None, stride=stride, padding=1)
        h = F.leaky_relu(_norm2d(h, sd, "model.%d" % (i + 1), "batch"), LRELU)
        i += 3
    return F.conv2d(h, sd["model.%d.weight" % i], sd["model.%d.bias" % i], stride=1, padding=1)


# --------------------------------------------------------------------------------------
# VGG19 feature extractor  (models/modules/architectures/perceptual.py:103-214)
# --------------------------------------------------------------------------------------
VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M",
             512, 512, 512, 512]
VGG19_NAMES = ["conv1_1", "conv1_2", "pool1", "conv2_1", "conv2_2", "pool2", "conv3_1", "conv3_2",
               "conv3_3", "conv3_4", "pool3", "conv4_1", "conv4_2", "conv4_3", "conv4_4", "pool4",
               "conv5_1", "conv5_2", "conv5_3", "conv5_4"]
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def vgg19_conv54(x, sd):
    """(x-mean)/std, then torchvision-cfg-E features up to conv5_4 *before* its ReLU
    (listen_list ['conv5_4'] -> features[:35], perceptual.py:124-169,201-214).  Keys:
    feature_net.convI_J.{weight,bias}."""
    mean = torch.tensor(IMAGENET_MEAN, dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=x.dtype).view(1, 3, 1, 1)
    x = (x - mean) / std
    last = VGG19_NAMES[-1]
    for name in VGG19_NAMES:
        if name.startswith("pool"):
            x = F.max_pool2d(x, 2, 2)
        else:
            x = _conv(x, sd, "feature_net." + name)
            if name != last:
                x = F.relu(x)
    return x


def vgg19_seeded_state(seed=1234):
    """Same He-normal fill as oracle/stubs/torchvision/models/vgg.py (the weights the reference
    run saw), keyed like FeatureExtractor.state_dict() minus the mean/std buffers."""
    g = torch.Generator().manual_seed(seed)
    sd, c = OrderedDict(), 3
    names = [n for n in VGG19_NAMES if n.startswith("conv")]
    chans = [v for v in VGG19_CFG if v != "M"]
    for n, v in zip(names, chans):
        sd["feature_net.%s.weight" % n] = torch.randn((v, c, 3, 3), generator=g) * math.sqrt(2.0 / (c * 9))
        sd["feature_net.%s.bias" % n] = torch.zeros(v)
        c = v
    return sd


# --------------------------------------------------------------------------------------
# Losses
# --------------------------------------------------------------------------------------
def bce_logits(x, target_is_real):
    """GANLoss 'vanilla' = BCEWithLogitsLoss(mean) against a constant label
    (models/modules/loss.py:85-86,104-137)."""
    t = torch.ones_like(x) if target_is_real else torch.zeros_like(x)
    return F.binary_cross_entropy_with_logits(x, t)


def ragan_g_loss(pred_fake, pred_real):
    """Relativistic generator loss (models/losses.py:428-433); pred_real is detached."""
    pred_real = pred_real.detach()
    return (bce_logits(pred_real - pred_fake.mean(), False) +
            bce_logits(pred_fake - pred_real.mean(), True)) / 2


def ragan_d_loss(pred_fake, pred_real):
    """Relativistic discriminator loss (models/losses.py:503-512): returns (l_real, l_fake)."""
    return (bce_logits(pred_real - pred_fake.mean(), True),
            bce_logits(pred_fake - pred_real.mean(), False))


# --------------------------------------------------------------------------------------
# Optimiser pieces
# --------------------------------------------------------------------------------------
def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (norm type 2): coef = max/(total+1e-6) clamped to 1
    (used through base_model.py:911-922 with grad_clip_value 0.1)."""
    total = torch.sqrt(sum((g.detach() ** 2).sum() for g in grads))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


class Adam:
    """torch.optim.Adam single-tensor algorithm, defaults lr 1e-4 betas (0.9, 0.999) eps 1e-8
    wd 0 (models/optimizers.py:130-132)."""

    def __init__(self, params, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
        self.params, self.lr, self.b1, self.b2, self.eps = list(params), lr, b1, b2, eps
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    def step(self, grads):
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        with torch.no_grad():
            for p, g, m, v in zip(self.params, grads, self.m, self.v):
                m.mul_(self.b1).add_(g, alpha=1 - self.b1)
                v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
                p.addcdiv_(m, denom, value=-self.lr / bc1)


# --------------------------------------------------------------------------------------
# The step
# --------------------------------------------------------------------------------------
def _is_param(k):
    return not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))


class OracleSRStep:
    """Restates SRModel.optimize_parameters (models/sr_model.py:195-267) for the ESRGAN recipe
    (options/sr/train_sr.yml:107-110,145-146,189-190): G forward; pix-l1*w + fea-vgg19-l1*w +
    RaGAN*w; backward; clip_grad_norm_(G, 0.1); Adam(G); D losses on detached fake; Adam(D).
    accumulations == 1 (virtual batch == batch)."""

    def __init__(self, g_state, d_state=None, vgg_state=None, *, arch="rrdb_net", nb=23, d_size=128,
                 d_nf=64, pixel_weight=1e-2, feature_weight=1.0, gan_weight=5e-3, lr=1e-4,
                 grad_clip=0.1, upsample_mode="upconv", d_arch="discriminator_vgg"):
        self.arch, self.nb, self.d_size, self.d_nf, self.d_arch = arch, nb, d_size, d_nf, d_arch
        self.upsample_mode = upsample_mode
        self.pw, self.fw, self.gw, self.clip = pixel_weight, feature_weight, gan_weight, grad_clip
        self.g = OrderedDict((k, v.clone().float()) for k, v in g_state.items())
        for v in self.g.values():
            v.requires_grad_(True)
        self.opt_g = Adam(self.g.values(), lr)
        self.d = None
        if d_state is not None and gan_weight:
            self.d = OrderedDict((k, v.clone()) for k, v in d_state.items())
            self.d_params = [v for k, v in self.d.items() if _is_param(k)]
            self.opt_d = Adam(self.d_params, lr)
        self.vgg = vgg_state if feature_weight else None
        self.log = OrderedDict()
        self.fake_H = None
        self.last_g_grads = None
        self.last_d_grads = None
        self.noise = None          # ESRGAN+ multiplier fields of the NEXT netG call (rrdbnet_forward's `noise`), set by the caller

    def netG(self, lr_img):
        if self.arch == "rrdb_net":
            return rrdbnet_forward(lr_img, self.g, self.nb, 4, self.upsample_mode, noise=self.noise)
        return srresnet_forward(lr_img, self.g, self.nb, 4)

    def netD(self, x):
        if self.d_arch == "unet":
            return unet_disc_forward(x, self.d)
        return disc_vgg_forward(x, self.d, self.d_size, self.d_nf, training=True)

    def step(self, LR, HR):
        log = self.log
        # ---- G (sr_model.py:199-252)
        if self.d is not None:
            for p in self.d_params:
                p.requires_grad_(False)
        fake = self.netG(LR)
        self.fake_H = fake
        total = 0
        if self.pw:
            l_pix = self.pw * F.l1_loss(fake, HR)                       # losses.py:37-39,858-859
            log["pix-l1"] = l_pix.item()
            total = total + l_pix
        if self.vgg is not None:
            fx = vgg19_conv54(fake, self.vgg)                              # losses.py:295-309
            fy = vgg19_conv54(HR.detach(), self.vgg)
            l_fea = 1 * (F.l1_loss(fx, fy) * 1 * self.fw)               # losses.py:303-309,846-851
            log["fea-vgg19-l1"] = l_fea.item()
            total = total + l_fea
        if self.d is not None:
            pf = self.netD(fake)                                           # losses.py:398-403
            pr = self.netD(HR)
            l_gan = self.gw * ragan_g_loss(pf, pr)
            log["l_g_gan"] = l_gan.item()
            total = total + l_gan
        g_params = list(self.g.values())
        grads = list(torch.autograd.grad(total, g_params))
        if self.clip:
            clip_grad_norm(grads, self.clip)                               # base_model.py:911-922
        self.last_g_grads = [g.clone() for g in grads]
        self.opt_g.step(grads)
        # ---- D (sr_model.py:253-267, base_model.py:852-883)
        if self.d is not None:
            self.HR = HR
            self._d_stage(fake)
        return OrderedDict(log)

    def step_chunked(self, LR, HR, chunk=1):
        """The same step (sr_model.py:195-267) for batches whose generator activations do not fit host memory (RRDBNet-23 at
        128 -> 512 saves ~3.5 GB per image for backward; BASELINE configs[1] is batch 16).  RRDBNet has no cross-sample
        coupling (RRDBNet_arch.py:48-60,150-163: convolutions and LeakyReLU only), so the chain rule may be cut at fake_H:
          (i)   fake_H = G(LR) chunk by chunk without a graph;
          (ii)  every generator loss on the FULL batch w.r.t. a leaf fake_H (the discriminator's BatchNorm statistics and the
                relativistic means, losses.py:432-433, couple the samples there) -> dL/dfake_H;
          (iii) per chunk: G forward with a graph, backward with that chunk's rows of dL/dfake_H injected, parameter
                gradients summed over chunks (the only difference from `step`: the summation order of the weight gradients);
          (iv)  clip + Adam(G), then the discriminator stage on the full batch exactly as in `step` (sr_model.py:253-267).
        Pinned against the REAL reference's batch-2 / batch-4 goldens by tests/test_oracle_golden.py."""
        log = self.log
        self.HR = HR
        N = LR.shape[0]
        cuts = [(a, min(a + chunk, N)) for a in range(0, N, chunk)]
        noise = self.noise

        def g_chunk(a, b):
            self.noise = None if noise is None else [m[a:b] for m in noise]
            return self.netG(LR[a:b])

        if self.d is not None:
            for p in self.d_params:
                p.requires_grad_(False)
        with torch.no_grad():
            fake = torch.cat([g_chunk(a, b) for a, b in cuts])
        self.fake_H = fake
        leaf = fake.clone().requires_grad_(True)
        total = 0
        if self.pw:
            l_pix = self.pw * F.l1_loss(leaf, HR)
            log["pix-l1"] = l_pix.item()
            total = total + l_pix
        if self.vgg is not None:
            fx = vgg19_conv54(leaf, self.vgg)
            fy = vgg19_conv54(HR.detach(), self.vgg)
            l_fea = 1 * (F.l1_loss(fx, fy) * 1 * self.fw)
            log["fea-vgg19-l1"] = l_fea.item()
            total = total + l_fea
            del fx, fy
        if self.d is not None:
            pf = self.netD(leaf)
            pr = self.netD(HR)
            l_gan = self.gw * ragan_g_loss(pf, pr)
            log["l_g_gan"] = l_gan.item()
            total = total + l_gan
        (dfake,) = torch.autograd.grad(total, [leaf])
        del total, leaf
        g_params = list(self.g.values())
        grads = [torch.zeros_like(p) for p in g_params]
        for a, b in cuts:
            out = g_chunk(a, b)
            part = torch.autograd.grad(out, g_params, grad_outputs=dfake[a:b])
            for acc, g in zip(grads, part):
                acc.add_(g)
            del out, part
        self.noise = noise
        if self.clip:
            clip_grad_norm(grads, self.clip)
        self.last_g_grads = [g.clone() for g in grads]
        self.opt_g.step(grads)
        if self.d is not None:
            self._d_stage(fake)
        return OrderedDict(log)

    def _d_stage(self, fake):
        """Discriminator stage (sr_model.py:253-267, base_model.py:852-883; losses.py:471-478,503-512)."""
        log = self.log
        for p in self.d_params:
            p.requires_grad_(True)
        pf = self.netD(fake.detach())
        pr = self.netD(self.HR)
        l_real, l_fake = ragan_d_loss(pf, pr)
        l_d = (l_fake + l_real) * 0.5
        log["l_d_real"], log["l_d_fake"] = l_real.item(), l_fake.item()
        log["D_real"], log["D_fake"] = pr.detach().mean().item(), pf.detach().mean().item()
        dgr = list(torch.autograd.grad(l_d, self.d_params))
        self.last_d_grads = [g.clone() for g in dgr]
        self.opt_d.step(dgr)

    def g_state(self):
        return OrderedDict((k, v.detach().clone()) for k, v in self.g.items())

    def d_state(self):
        return OrderedDict((k, v.detach().clone()) for k, v in self.d.items())


def psnr_reference(sr, hr, scale=4):
    """tensor2np (x255, clip, round, uint8; dataops/common.py:560-566) on the first image, crop
    `scale` px (utils/metrics.py:59-60, called with crop_size=opt['scale'] train.py:372),
    20*log10(255/sqrt(mse)) (utils/metrics.py:110-126)."""
    def to_u8(t):
        t = t.detach().float().cpu()
        if t.dim() == 4:
            t = t[0]
        return (t * 255.0).clamp(0, 255).round().to(torch.uint8)
    a, b = to_u8(sr).double(), to_u8(hr).double()
    a = a[:, scale:-scale, scale:-scale]
    b = b[:, scale:-scale, scale:-scale]
    mse = ((a - b) ** 2).mean().item()
    if mse == 0:
        return float("inf")
    return 20.0 * math.log10(255.0 / math.sqrt(mse))
