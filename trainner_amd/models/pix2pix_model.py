"""Pix2PixModel on the MI355X engine -- the drop-in for codes/models/pix2pix_model.py.

Same constructor sequence (:38-122), `feed_data` (:124-138), `forward` (:140-142), `backward_D` (:144-148: conditional
`backward_D_Basic` on the (real_A, .) pairs), `backward_G` (:150-177: conditional GAN loss + the generator loss list) and
**`optimize_parameters`** (:179-235: G(A) once, then the D step, then the G step against the updated D), and the log /
visual accessors (:237-248), so codes/train.py drives it unchanged.  netG (ResnetGenerator) / netD (PatchGAN over the
6-channel (A, B) concatenation) are HIP-engine networks, the GAN criterion (`gan_opt.form: standard`, vanilla or lsgan) and
L1 are HIP kernels, Adam runs on the flat buffers, gradients are exchanged with RCCL when WORLD_SIZE > 1.
"""
import logging
from collections import OrderedDict

from . import losses, networks
from .base_model import BaseModel, LazyLog, training_step

logger = logging.getLogger("base")


class Pix2PixModel(BaseModel):
    def __init__(self, opt):
        super().__init__(opt)
        train_opt = opt["train"]
        self.visual_names = ["real_A", "fake_B", "real_B"]
        self.model_names = ["G"]
        self.netG = networks.define_G(opt).to(self.device)
        if self.is_train:
            self.netG.train()
            opt_G_nets, opt_D_nets = [self.netG], []
            if train_opt["gan_weight"]:
                self.model_names.append("D")
                self.netD = networks.define_D(opt).to(self.device)      # in_nc = input_nc + output_nc (train_pix2pix.yml:107)
                self.netD.train()
                opt_D_nets.append(self.netD)
            self.setup_atg()
        self.load()
        if self.is_train:
            self.setup_batchaug()
            self.setup_fs()
            self.generatorlosses = losses.GeneratorLoss(opt, self.device)
            self.generatorlosses.dp_group = self.dp if self.dp.active else None
            self.setup_gan(conditional=True)
            if self.cri_gan:
                self.setup_freezeD()
            self.setup_optimizers(opt_G_nets, opt_D_nets, init_setup=True)
            self.setup_schedulers()
            self.optimizer_G.zero_grad()
            if self.cri_gan:
                self.optimizer_D.zero_grad()
            self.log_dict = LazyLog()
            self.setup_swa()
            self.setup_virtual_batch()
            self.setup_amp()
            self.sync_replicas()
        self.print_network(verbose=False)

    def feed_data(self, data):
        self.real_A = self._shard(data["A"]).to(self.device, non_blocking=True)
        self.real_B = self._shard(data["B"]).to(self.device, non_blocking=True)
        self.image_paths = data.get("A_path")

    def forward(self):
        self.fake_B = self.netG(self.real_A)

    def backward_D(self):
        self._arm_bucket_schedule([self.netD], passes=2)
        self.log_dict = self.backward_D_Basic(self.netD, self.real_B, self.fake_B, self.log_dict, self.real_A)

    def backward_G(self):
        l_g_total = 0
        if self.cri_gan:
            l_g_gan = self.adversarial(self.fake_B, condition=self.real_A, netD=self.netD, stage="generator", fsfilter=self.f_high)
            self.log_dict["l_g_gan"] = self.adversarial._logged(l_g_gan)
            l_g_total = l_g_total + (l_g_gan if self.accumulations == 1 else l_g_gan / self.accumulations)
        loss_results, self.log_dict = self.generatorlosses(self.fake_B, self.real_B, self.log_dict, self.f_low)
        l_g = sum(loss_results)
        l_g_total = l_g_total + (l_g if self.accumulations == 1 else l_g / self.accumulations)
        self._arm_bucket_schedule([self.netG])
        self.calc_gradients(l_g_total)

    @training_step
    def optimize_parameters(self, step):
        eff_step = step / self.accumulations
        self.forward()
        if self.cri_gan:
            self.requires_grad(self.netD, flag=True)
            if isinstance(self.feature_loc, int):
                for loc in range(self.feature_loc):
                    self.requires_grad(self.netD, False, target_layer=loc, net_type="D")
            self.backward_D()
            self.optimizer_step(step, self.optimizer_D, "D")
        if (self.cri_gan is not True) or (eff_step % self.D_update_ratio == 0 and eff_step > self.D_init_iters):
            if self.cri_gan:
                self.requires_grad(self.netD, flag=False, net_type="D")
            self.backward_G()
            self.optimizer_step(step, self.optimizer_G, "G")

    def get_current_log(self):
        log = self.log_dict.materialize() if isinstance(self.log_dict, LazyLog) else OrderedDict(self.log_dict)
        self.check_engine_errors()
        return log

    def get_current_visuals(self):
        out = OrderedDict()
        for name in self.visual_names:
            out[name] = getattr(self, name).detach()[0].float().cpu()
        return out
