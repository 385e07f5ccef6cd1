"""Loss builders of the SR training path, backed by HIP kernels.

Keeps the reference's surface (codes/models/losses.py): `get_loss_fn` -> {'name','weight','function'},
`PerceptualLoss` (:220-340), `Adversarial` (:343-604) and `GeneratorLoss` (:607-962) with the same
option keys, loss names (`pix-l1`, `fea-vgg19-l1`) and weighting order, restricted to the branches the
ESRGAN recipe uses (options/sr/train_sr.yml:107-110,145-146): L1 pixel loss, VGG19 conv5_4 L1
perceptual loss, vanilla relativistic GAN.  Anything else raises NotImplementedError (no silent
fallback to eager PyTorch).
"""
import torch
import torch.nn as nn

from .. import hip, ops
from . import networks


# ----------------------------------------------------------------------------------------------
# HIP-backed criteria
# ----------------------------------------------------------------------------------------------
def _same_dense_layout(a, b):
    return a.shape == b.shape and a.stride() == b.stride() and (
        a.is_contiguous() or (a.dim() == 4 and a.permute(0, 2, 3, 1).is_contiguous()))


class _L1MeanFn(torch.autograd.Function):
    """mean(|a - b|): nn.L1Loss(reduction='mean') (losses.py:37-39).  b carries no gradient."""

    @staticmethod
    def forward(ctx, a, b):
        hip.require_device(a)
        if not _same_dense_layout(a, b):
            raise hip.HipEngineError("L1: operands must share one dense layout")
        out = torch.empty((), dtype=torch.float32, device=a.device)
        ops.l1_mean_fwd(a, b, 1.0, out)
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = torch.empty_like(a)          # preserves a's (possibly channels-last) strides
        ops.l1_mean_bwd(a, b, 1.0, g.contiguous(), ga)
        return ga, None


class L1Loss(nn.Module):
    def __init__(self, reduction="mean"):
        super().__init__()
        if reduction != "mean":
            raise NotImplementedError("only reduction='mean' is implemented by the HIP engine")

    def forward(self, x, y):
        return _L1MeanFn.apply(x, y.detach())


class _RaGANFn(torch.autograd.Function):
    """Relativistic average BCE-with-logits (GANLoss 'vanilla', modules/loss.py:85-86,120-137) in
    the generator (losses.py:428-433) or discriminator (losses.py:503-512) form.  Returns a 5-vector:
    [weight*(l1+l2)/2, l1, l2, mean(pred_real), mean(pred_fake)].  With a DataParallel group the two
    batch means and the two coupling sums are all-reduced between the kernel phases, so every rank
    sees the GLOBAL-batch relativistic means the reference computes on GPU 0."""

    @staticmethod
    def forward(ctx, pred_fake, pred_real, stage, weight, group):
        hip.require_device(pred_fake)
        pf, pr = pred_fake.contiguous().view(-1), pred_real.contiguous().view(-1)
        dev = pf.device
        sums = torch.empty(8, dtype=torch.float32, device=dev)
        out = torch.empty(5, dtype=torch.float32, device=dev)
        gf = torch.empty_like(pf)
        gr = torch.empty_like(pr)
        ops.ragan_phase_a(pf, pr, sums)
        if group is not None:
            group.all_reduce_sum(sums[0:3])
        ops.ragan_phase_b(pf, pr, stage, sums)
        if group is not None:
            group.all_reduce_sum(sums[3:7])
        ops.ragan_phase_c(pf, pr, stage, weight, sums, out, gf, gr)
        ctx.save_for_backward(gf, gr)
        ctx.shapes = (pred_fake.shape, pred_real.shape)
        ctx.world = 1 if group is None else group.world_size
        return out

    @staticmethod
    def backward(ctx, g):
        gf, gr = ctx.saved_tensors
        g0 = g[0:1].contiguous()
        if ctx.world > 1:
            # every rank back-propagates the global-mean loss; gradient averaging over ranks (dp.py)
            # then divides by world, so pre-multiply to keep d(global loss)/d(local sample) exact
            g0 = g0 * float(ctx.world)
        of = torch.empty_like(gf)
        ops.scale_by(of, gf, g0)
        orr = None
        if ctx.needs_input_grad[1]:
            orr = torch.empty_like(gr)
            ops.scale_by(orr, gr, g0)
            orr = orr.view(ctx.shapes[1])
        return of.view(ctx.shapes[0]), orr, None, None, None


class _GanLabelFn(torch.autograd.Function):
    """GANLoss against a constant label (modules/loss.py:85-88,112-137): mean BCE-with-logits ('vanilla', kind 0) or mean
    squared error ('lsgan', kind 1) of the discriminator's logit map against `target` (1.0 real / 0.0 fake).  One launch
    produces the loss and d loss / d pred (tnr_gan_loss)."""

    @staticmethod
    def forward(ctx, pred, kind, target):
        hip.require_device(pred)
        p = pred.contiguous().view(-1)
        out = torch.empty(1, dtype=torch.float32, device=p.device)
        grad = torch.empty_like(p) if ctx.needs_input_grad[0] else None
        ops.gan_loss(p, kind, target, out, grad)
        ctx.save_for_backward(grad)
        ctx.shape = pred.shape
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        of = torch.empty_like(grad)
        ops.scale_by(of, grad, g.reshape(1).contiguous())
        return of.view(ctx.shape), None, None


class _CatChannelsFn(torch.autograd.Function):
    """torch.cat((a, b), 1) of two NCHW image batches for the conditional discriminator input (losses.py:445-455,
    522-530): two strided device copies into one buffer; backward hands each operand its channel slice."""

    @staticmethod
    def forward(ctx, a, b):
        out = torch.empty((a.shape[0], a.shape[1] + b.shape[1]) + tuple(a.shape[2:]), dtype=a.dtype, device=a.device)
        out[:, :a.shape[1]].copy_(a)
        out[:, a.shape[1]:].copy_(b)
        ctx.ca = a.shape[1]
        return out

    @staticmethod
    def backward(ctx, g):
        ga = g[:, :ctx.ca].contiguous() if ctx.needs_input_grad[0] else None
        gb = g[:, ctx.ca:].contiguous() if ctx.needs_input_grad[1] else None
        return ga, gb


# ----------------------------------------------------------------------------------------------
# builders
# ----------------------------------------------------------------------------------------------
def get_loss_fn(loss_type=None, weight=0, recurrent=False, reduction="mean", network=None, device="cuda", opt=None,
                allow_featnets=True):
    """Same contract as the reference (losses.py:23-171): returns the criterion itself when
    `recurrent`, else {'name', 'weight', 'function'}."""
    if loss_type in ("L1", "l1"):
        loss_function = L1Loss(reduction=reduction)
        loss_type = "pix-{}".format(loss_type)
    elif loss_type is not None and loss_type.find("fea") >= 0:
        parts = loss_type.split("-")
        if parts[1] == "lpips":
            raise NotImplementedError("LPIPS is outside the SR hot path of the HIP engine")
        fea_loss_f = get_loss_fn(parts[2], recurrent=True, reduction="mean", device=device)
        network = networks.define_F(opt).to(device)
        loss_function = PerceptualLoss(criterion=fea_loss_f, network=network, opt=opt)
    else:
        raise NotImplementedError("Loss type [{}] is not implemented by the HIP engine".format(loss_type))
    if recurrent:
        return loss_function.to(device)
    return {"name": loss_type, "weight": weight, "function": loss_function.to(device)}


def check_loss_names(feature_criterion=None, feature_network=None, **_unused):
    """losses.py:174-217 for the feature-loss name only."""
    if feature_criterion and feature_network:
        return "fea-{}-{}".format(feature_network.lower(), feature_criterion.lower())
    return None


class PerceptualLoss(nn.Module):
    """VGG feature (perceptual) loss; style loss / random flips are not on the path (losses.py:220-340)."""

    def __init__(self, criterion=None, network=None, opt=None):
        super().__init__()
        self.criterion, self.network = criterion, network
        w_l_p = {"conv5_4": 1}
        self.perceptual_weight, self.style_weight = 1.0, 0.0
        if opt:
            train_opt = opt["train"]
            self.perceptual_weight = train_opt.get("feature_weight", 0) or 0
            self.style_weight = train_opt.get("style_weight", 0) or 0
            perc_opts = train_opt.get("perceptual_opt")
            if perc_opts:
                w_l_p = perc_opts.get("perceptual_layers", {"conv5_4": 1})
                if perc_opts.get("rotations") or perc_opts.get("flips") or perc_opts.get("style_layers"):
                    raise NotImplementedError("perceptual_opt rotations/flips/style are not implemented by the HIP engine")
        if self.style_weight > 0:
            raise NotImplementedError("style loss is not implemented by the HIP engine")
        self.w_l_p = w_l_p

    def forward(self, x, y):
        fea_x = self.network(x)
        with torch.no_grad():
            fea_y = self.network(y.detach())
        percep_loss = None
        if self.perceptual_weight > 0:
            percep_loss = 0
            for k in self.w_l_p.keys():
                percep_loss = percep_loss + self.criterion(fea_x[k], fea_y[k]) * self.w_l_p[k]
            percep_loss = percep_loss * self.perceptual_weight
        return percep_loss, None


class Adversarial(nn.Module):
    """Discriminator-driven losses (losses.py:343-604) for single-scale discriminators without feature maps:
    `gan_opt.form` relativistic (vanilla GAN: the ESRGAN recipe) or standard (vanilla / lsgan: Pix2Pix, CycleGAN), and the
    conditional formulation (Pix2Pix: D sees the (condition, image) channel concatenation, losses.py:445-455,522-530)."""

    _KINDS = {"vanilla": 0, "lsgan": 1}

    def __init__(self, train_opt=None, device="cpu", diffaug=False, dapolicy="", conditional=False):
        super().__init__()
        if diffaug or train_opt.get("gan_featmaps"):
            raise NotImplementedError("diffaug / feature-map GAN options are not implemented by the HIP engine")
        self.device = device
        self.conditional = bool(conditional)
        self.gan_type = train_opt["gan_type"]
        if self.gan_type not in self._KINDS:
            raise NotImplementedError("GAN type [{}] is not implemented by the HIP engine".format(self.gan_type))
        self.l_gan_w = train_opt["gan_weight"]
        self.form = (train_opt.get("gan_opt") or {}).get("form", "relativistic")
        if self.form not in ("relativistic", "standard"):
            raise NotImplementedError("GAN form [{}] is not implemented by the HIP engine".format(self.form))
        if self.form == "relativistic" and self.gan_type != "vanilla":
            raise NotImplementedError("the relativistic form is implemented for gan_type vanilla only")
        self.dp_group = None        # set by the model when running data-parallel

    def _label_loss(self, pred, target_is_real):
        return _GanLabelFn.apply(pred, self._KINDS[self.gan_type], 1.0 if target_is_real else 0.0)

    def _logged(self, t):
        """Logged scalars are global-batch means under data parallelism (equal shards), as the reference's gathered batch."""
        t = t.detach()
        return self.dp_group.mean_scalar(t) if self.dp_group is not None else t

    def forward(self, fake, real=None, condition=None, netD=None, stage="discriminator", fsfilter=None):
        if fsfilter is not None:
            raise NotImplementedError("frequency separation is not implemented by the HIP engine")
        if self.conditional:
            # like the reference's dispatch (losses.py:590-604) the second positional argument is the condition in the
            # generator stage (pix2pix_model.py:152-154 passes it by keyword)
            if condition is None:
                raise ValueError("conditional GAN: no condition image was given")
            fake = _CatChannelsFn.apply(condition, fake)
            if real is not None:
                real = _CatChannelsFn.apply(condition, real)
        if stage == "generator":
            pred_g_fake = netD(fake)
            if self.form == "standard":                # D(real) is not needed (losses.py:395-403,424-426)
                return self.l_gan_w * self._label_loss(pred_g_fake, True)
            if real is None:
                # the reference reaches netD(None) here (losses.py:401-403: conv2d on None => TypeError) -- e.g. its shipped
                # options/i2i/train_pix2pix.yml, which has no `gan_opt` and calls the generator stage without `real`
                raise TypeError("the relativistic GAN form (the default when `train.gan_opt.form` is not given) needs the real "
                                "image in the generator stage and none was passed (the reference fails at the same place): "
                                "set train.gan_opt.form: standard")
            with torch.no_grad():
                pred_g_real = netD(real)          # detached in the reference (losses.py:430)
            res = _RaGANFn.apply(pred_g_fake, pred_g_real, 0, self.l_gan_w, self.dp_group)
            return res[0]
        pred_d_fake = netD(fake.detach())
        pred_d_real = netD(real)
        if self.form == "standard":                    # losses.py:497-499,514-523
            l_d_fake = self._label_loss(pred_d_fake, False)
            l_d_real = self._label_loss(pred_d_real, True)
            l_d_total = (l_d_fake + l_d_real) * 0.5
            gan_logs = {"l_d_real": self._logged(l_d_real), "l_d_fake": self._logged(l_d_fake),
                        "D_real": self._logged(ops_mean(pred_d_real)), "D_fake": self._logged(ops_mean(pred_d_fake))}
            return l_d_total, gan_logs
        res = _RaGANFn.apply(pred_d_fake, pred_d_real, 1, 1.0, self.dp_group)
        # kept on device: the model's log dict materialises lazily (one sync instead of four .item())
        gan_logs = {"l_d_real": res[1].detach(), "l_d_fake": res[2].detach(),
                    "D_real": res[3].detach(), "D_fake": res[4].detach()}
        return res[0], gan_logs


def ops_mean(pred):
    """torch.mean(pred.detach()) of a logit map for the D_real / D_fake log entries (losses.py:519-520)."""
    p = pred.detach().contiguous().view(-1)
    out = torch.empty(1, dtype=torch.float32, device=p.device)
    ops.gan_loss(p, 2, 0.0, out, None)
    return out[0]


class GeneratorLoss(nn.Module):
    """Weighted list of generator losses (losses.py:607-962): pixel then feature, same order/names."""

    _UNSUPPORTED = ("hfen_weight", "tv_weight", "color_weight", "avg_weight", "ms_weight", "spl_weight", "of_weight",
                    "style_weight", "lpips_weight", "cx_weight", "grad_weight", "ssim_weight", "fft_weight",
                    "fdpl_weight", "range_weight")

    def __init__(self, opt=None, device="cpu", allow_featnets=True):
        super().__init__()
        train_opt = opt["train"]
        for k in self._UNSUPPORTED:
            if train_opt.get(k):
                raise NotImplementedError("loss option '{}' is outside the SR hot path of the HIP engine".format(k))
        pixel_weight = train_opt.get("pixel_weight", 0) or 0
        pixel_criterion = train_opt.get("pixel_criterion", None)
        self.loss_list = []
        if pixel_weight > 0 and pixel_criterion:
            self.loss_list.append(get_loss_fn(pixel_criterion, pixel_weight, device=device))
        feature_weight = (train_opt.get("feature_weight", 0) or 0) if allow_featnets else 0
        feat_opts = train_opt.get("perceptual_opt")
        feature_network = (feat_opts or {}).get("feature_network", None) or train_opt.get("feature_network", "vgg19") or "vgg19"
        feature_criterion = check_loss_names(feature_criterion=train_opt.get("feature_criterion"),
                                             feature_network=feature_network)
        if feature_weight > 0 and feature_criterion:
            self.loss_list.append(get_loss_fn(feature_criterion, 1, opt=opt, device=device))
            self.cri_fea = True
        else:
            self.cri_fea = None
        self.precise_loss_list = []
        self.dp_group = None        # set by SRModel when running data-parallel

    def forward(self, sr, hr, log_dict, fsfilter=None, selector=None, precise=False):
        if fsfilter is not None or selector:
            raise NotImplementedError("frequency separation / loss selectors are not implemented by the HIP engine")
        if precise:
            return [], log_dict
        results = []
        for l in self.loss_list:
            if "fea-vgg" in l["name"]:
                percep_loss, _ = l["function"](sr, hr)
                effective = l["weight"] * percep_loss
            else:
                effective = l["weight"] * l["function"](sr, hr)
            results.append(effective)
            # under data parallelism the logged value is the global-batch mean, as the reference computes it on the
            # gathered batch (the gradient uses the local mean: averaging over ranks makes it the global one)
            log_dict[l["name"]] = self.dp_group.mean_scalar(effective) if self.dp_group is not None else effective.detach()
        return results, log_dict
