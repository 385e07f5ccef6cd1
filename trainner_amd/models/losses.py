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
    """Discriminator-driven losses (losses.py:343-604): vanilla GAN, relativistic or standard form."""

    def __init__(self, train_opt=None, device="cpu", diffaug=False, dapolicy="", conditional=False):
        super().__init__()
        if diffaug or conditional or train_opt.get("gan_featmaps"):
            raise NotImplementedError("diffaug / conditional / feature-map GAN options are not implemented by the HIP engine")
        self.device = device
        self.gan_type = train_opt["gan_type"]
        if self.gan_type != "vanilla":
            raise NotImplementedError("GAN type [{}] is not implemented by the HIP engine".format(self.gan_type))
        self.l_gan_w = train_opt["gan_weight"]
        self.form = (train_opt.get("gan_opt") or {}).get("form", "relativistic")
        if self.form != "relativistic":
            raise NotImplementedError("GAN form [{}] is not implemented by the HIP engine".format(self.form))
        self.dp_group = None        # set by SRModel when running data-parallel

    def forward(self, fake, real=None, condition=None, netD=None, stage="discriminator", fsfilter=None):
        if fsfilter is not None:
            raise NotImplementedError("frequency separation is not implemented by the HIP engine")
        if stage == "generator":
            pred_g_fake = netD(fake)
            with torch.no_grad():
                pred_g_real = netD(real)          # detached in the reference (losses.py:430)
            res = _RaGANFn.apply(pred_g_fake, pred_g_real, 0, self.l_gan_w, self.dp_group)
            return res[0]
        pred_d_fake = netD(fake.detach())
        pred_d_real = netD(real)
        res = _RaGANFn.apply(pred_d_fake, pred_d_real, 1, 1.0, self.dp_group)
        # kept on device: SRModel's log dict materialises lazily (one sync instead of four .item())
        gan_logs = {"l_d_real": res[1].detach(), "l_d_fake": res[2].detach(),
                    "D_real": res[3].detach(), "D_fake": res[4].detach()}
        return res[0], gan_logs


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
