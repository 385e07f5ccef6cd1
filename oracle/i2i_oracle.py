"""CPU restatement of the reference's image-to-image training steps (the ORACLE for Pix2Pix / CycleGAN).

TEST INFRASTRUCTURE ONLY.  Nothing under trainner_amd/ may import this module.  Plain fp32 PyTorch on the CPU, functional
over state_dicts with the reference's own keys; PINNED against the real reference: tests/test_oracle_golden.py compares it
with tests/golden/{pix2pix,cyclegan}_*.pt, which oracle/make_golden_i2i.py produced from /root/reference itself.
Citations are relative to /root/reference/codes.
"""
import random
from collections import OrderedDict

import torch
import torch.nn.functional as F

from .sr_oracle import Adam, _is_param, patchgan_forward, resnet_generator_forward, unet_generator_forward


def gan_label_loss(pred, target_is_real, gan_type):
    """GANLoss against a constant label (models/modules/loss.py:85-88,112-137): BCE-with-logits ('vanilla') or MSE ('lsgan')."""
    t = torch.ones_like(pred) if target_is_real else torch.zeros_like(pred)
    return F.binary_cross_entropy_with_logits(pred, t) if gan_type == "vanilla" else F.mse_loss(pred, t)


class _Net:
    def __init__(self, state, lr_unused=None):
        self.sd = OrderedDict((k, v.clone()) for k, v in state.items())
        self.params = [v for k, v in self.sd.items() if _is_param(k)]
        for p in self.params:
            p.requires_grad_(True)

    def state(self):
        return OrderedDict((k, v.detach().clone()) for k, v in self.sd.items())


class OraclePool:
    """utils/image_pool.py:5-58 (same `random` draw sequence)."""

    def __init__(self, size):
        self.size, self.images = size, []

    def query(self, images):
        if self.size == 0:
            return images
        out = []
        for img in images:
            img = img.detach().unsqueeze(0)
            if len(self.images) < self.size:
                self.images.append(img)
                out.append(img)
            elif random.uniform(0, 1) > 0.5:
                j = random.randint(0, self.size - 1)
                out.append(self.images[j].clone())
                self.images[j] = img
            else:
                out.append(img)
        return torch.cat(out, 0)


class _I2IBase:
    def __init__(self, n_blocks, norm, gan_type, pixel_weight, lr, beta1):
        self.nb, self.norm, self.gan_type, self.pw, self.lr, self.b1 = n_blocks, norm, gan_type, pixel_weight, lr, beta1
        self.log = OrderedDict()

    arch, num_downs = "resnet_net", None          # generator: ResnetGenerator (n_blocks) or UnetGenerator (num_downs)
    form = "standard"                             # train.gan_opt.form; 'relativistic' when the recipe has no gan_opt (losses.py:366-369)

    def g_gan_loss(self, netD, fake, real):
        """losses.py:395-403,424-431: standard = label loss on D(fake); relativistic = the average-relativistic pair, D(real) detached."""
        pf = self.D(netD, fake)
        if self.form == "standard":
            return gan_label_loss(pf, True, self.gan_type)
        pr = self.D(netD, real).detach()
        return (gan_label_loss(pr - pf.mean(), False, self.gan_type) + gan_label_loss(pf - pr.mean(), True, self.gan_type)) / 2

    def G(self, net, x):
        if self.arch == "unet_net":
            return unet_generator_forward(x, net.sd, self.num_downs, self.norm)
        return resnet_generator_forward(x, net.sd, self.nb, self.norm)

    def D(self, net, x):
        return patchgan_forward(x, net.sd, 3)

    def d_step_loss(self, netD, real, fake, log):
        """base_model.py:852-883 + losses.py:471-478,497-523: D(fake.detach()) first, then D(real); standard labels or the
        relativistic pair (:501-506)."""
        pf = self.D(netD, fake.detach())
        pr = self.D(netD, real)
        if self.form == "standard":
            l_fake, l_real = gan_label_loss(pf, False, self.gan_type), gan_label_loss(pr, True, self.gan_type)
        else:
            l_real = gan_label_loss(pr - pf.mean(), True, self.gan_type)
            l_fake = gan_label_loss(pf - pr.mean(), False, self.gan_type)
        log["l_d_real"], log["l_d_fake"] = l_real.item(), l_fake.item()
        log["D_real"], log["D_fake"] = pr.detach().mean().item(), pf.detach().mean().item()
        return (l_fake + l_real) * 0.5

    @staticmethod
    def _freeze(net, flag):
        for p in net.params:
            p.requires_grad_(flag)


class OraclePix2PixStep(_I2IBase):
    """Pix2PixModel.optimize_parameters (models/pix2pix_model.py:179-235) for the recipe options/i2i/train_pix2pix.yml with
    the ResNet generator: fake_B = G(A); D step on the conditional pairs (A, fake_B.detach()) / (A, B); Adam(D); G step:
    gan_weight * GAN(D(A, fake_B), real) + pixel_weight * L1(fake_B, B) against the UPDATED D; Adam(G)."""

    def __init__(self, g_state, d_state, *, n_blocks, norm="instance", gan_type="vanilla", pixel_weight=100.0, gan_weight=1.0,
                 lr=2e-4, beta1=0.5):
        super().__init__(n_blocks, norm, gan_type, pixel_weight, lr, beta1)
        self.gw = gan_weight
        self.g, self.d = _Net(g_state), _Net(d_state)
        self.opt_g = Adam(self.g.params, lr, b1=beta1)
        self.opt_d = Adam(self.d.params, lr, b1=beta1)
        self.fake_B = None

    def step(self, A, B):
        log = self.log
        fake = self.G(self.g, A)
        self.fake_B = fake
        self._freeze(self.d, True)
        l_d = self.d_step_loss(self.d, torch.cat((A, B), 1), torch.cat((A, fake), 1), log)
        self.last_d_grads = list(torch.autograd.grad(l_d, self.d.params))
        self.opt_d.step(self.last_d_grads)
        self._freeze(self.d, False)
        l_gan = self.gw * gan_label_loss(self.D(self.d, torch.cat((A, fake), 1)), True, self.gan_type)   # losses.py:445-455,424-426
        log["l_g_gan"] = l_gan.item()
        l_pix = self.pw * F.l1_loss(fake, B)
        log["pix-l1"] = l_pix.item()
        self.last_g_grads = list(torch.autograd.grad(l_gan + l_pix, self.g.params))
        self.opt_g.step(self.last_g_grads)
        return OrderedDict(log)


class OracleCycleGANStep(_I2IBase):
    """CycleGANModel.optimize_parameters (models/cyclegan_model.py:309-370): forward (:193-198), backward_G (:212-307: identity
    terms x lambda_identity, GAN terms, cycle terms; one Adam over G_A + G_B), then backward_D_A / backward_D_B on pooled
    fakes (:200-210) and one Adam over D_A + D_B.  Logs: per-direction dicts folded into `log` with _A / _B suffixes inside
    backward_G, i.e. BEFORE the D step of the same iteration (D entries show up one step late, like the reference)."""

    def __init__(self, ga, gb, da, db, *, n_blocks, norm="instance", gan_type="vanilla", pixel_weight=10.0, gan_weight=1.0,
                 lambda_identity=0.5, pool_size=0, lr=2e-4, beta1=0.5):
        super().__init__(n_blocks, norm, gan_type, pixel_weight, lr, beta1)
        self.gw, self.idt = gan_weight, lambda_identity
        self.ga, self.gb, self.da, self.db = _Net(ga), _Net(gb), _Net(da), _Net(db)
        self.opt_g = Adam(self.ga.params + self.gb.params, lr, b1=beta1)
        self.opt_d = Adam(self.da.params + self.db.params, lr, b1=beta1)
        self.pool_A, self.pool_B = OraclePool(pool_size), OraclePool(pool_size)
        self.log_A, self.log_B = OrderedDict(), OrderedDict()

    def step(self, A, B):
        fake_B = self.G(self.ga, A)
        rec_A = self.G(self.gb, fake_B)
        fake_A = self.G(self.gb, B)
        rec_B = self.G(self.ga, fake_A)
        self.fake_B, self.fake_A, self.rec_A, self.rec_B = fake_B, fake_A, rec_A, rec_B
        self._freeze(self.da, False)
        self._freeze(self.db, False)
        total = 0
        if self.idt and self.idt > 0:
            idt_A, idt_B = self.G(self.ga, B), self.G(self.gb, A)
            l = self.pw * F.l1_loss(idt_A, B)
            self.log_A["pix-l1_idt"] = l.item()
            total = total + l * self.idt
            l = self.pw * F.l1_loss(idt_B, A)
            self.log_B["pix-l1_idt"] = l.item()
            total = total + l * self.idt
        l = self.gw * self.g_gan_loss(self.da, fake_B, A)          # (fake_B, real_A): cyclegan_model.py:241-243
        self.log_A["l_g_gan"] = l.item()
        total = total + l
        l = self.gw * self.g_gan_loss(self.db, fake_A, B)          # (fake_A, real_B): :248-250
        self.log_B["l_g_gan"] = l.item()
        total = total + l
        l = self.pw * F.l1_loss(rec_A, A)
        self.log_A["pix-l1"] = l.item()
        total = total + l
        l = self.pw * F.l1_loss(rec_B, B)
        self.log_B["pix-l1"] = l.item()
        total = total + l
        self.last_g_grads = list(torch.autograd.grad(total, self.ga.params + self.gb.params))
        for k, v in self.log_A.items():
            self.log["%s_A" % k] = v
        for k, v in self.log_B.items():
            self.log["%s_B" % k] = v
        self.opt_g.step(self.last_g_grads)
        self._freeze(self.da, True)
        self._freeze(self.db, True)
        l_da = self.d_step_loss(self.da, B, self.pool_B.query(fake_B), self.log_A)
        g_da = list(torch.autograd.grad(l_da, self.da.params))
        l_db = self.d_step_loss(self.db, A, self.pool_A.query(fake_A), self.log_B)
        g_db = list(torch.autograd.grad(l_db, self.db.params))
        self.last_d_grads = g_da + g_db
        self.opt_d.step(self.last_d_grads)
        return OrderedDict(self.log)
