"""CycleGANModel on the MI355X engine -- the drop-in for codes/models/cyclegan_model.py.

Same constructor sequence (:53-175: two generators G_A (A -> B), G_B (B -> A), two discriminators D_A (G_A(A) vs B), D_B
(G_B(B) vs A), two image pools, one optimizer over both generators and one over both discriminators), `feed_data`
(:177-191), `forward` (:193-198: fake_B, rec_A, fake_A, rec_B), `backward_D_A/B` (:200-210: pool query + `backward_D_Basic`),
`backward_G` (:212-307: identity terms weighted by `lambda_identity`, two GAN terms, two cycle terms through the generator
loss list; per-direction logs folded into `log_dict` with `_A` / `_B` suffixes) and **`optimize_parameters`** (:309-370: the
G step first, then both D steps on pooled fakes), so codes/train.py drives it unchanged.  Every network is a HIP-engine
network (ResnetGenerator / PatchGAN); the generators run three times per step (translation, cycle, identity) and their
parameter gradients accumulate in the flat gradient buffers that Adam then consumes in one launch per network.
"""
import logging
from collections import OrderedDict

from ..utils.image_pool import ImagePool
from . import losses, networks
from .base_model import BaseModel, LazyLog, training_step

logger = logging.getLogger("base")


class CycleGANModel(BaseModel):
    def __init__(self, opt):
        super().__init__(opt)
        train_opt = opt["train"]
        self.lambda_idt = train_opt["lambda_identity"] if self.is_train else None
        self.use_idt = bool(self.is_train and self.lambda_idt and self.lambda_idt > 0.0)
        visual_names_A, visual_names_B = ["real_A", "fake_B", "rec_A"], ["real_B", "fake_A", "rec_B"]
        if self.use_idt:
            visual_names_A.append("idt_B")
            visual_names_B.append("idt_A")
        self.visual_names = visual_names_A + visual_names_B
        self.model_names = ["G_A"]
        self.netG_A = networks.define_G(opt).to(self.device)
        if self.is_train:
            self.model_names.append("G_B")
            self.netG_B = networks.define_G(opt).to(self.device)
            self.netG_A.train()
            self.netG_B.train()
            opt_G_nets, opt_D_nets = [self.netG_A, self.netG_B], []
            if train_opt["gan_weight"]:
                self.model_names += ["D_A", "D_B"]
                self.netD_A = networks.define_D(opt).to(self.device)
                self.netD_B = networks.define_D(opt).to(self.device)
                self.netD_A.train()
                self.netD_B.train()
                opt_D_nets += [self.netD_A, self.netD_B]
            self.setup_atg()
        self.load()
        if self.is_train:
            if self.use_idt:
                assert opt["input_nc"] == opt["output_nc"]
            self.fake_A_pool = ImagePool(opt["pool_size"])
            self.fake_B_pool = ImagePool(opt["pool_size"])
            self.setup_batchaug()
            self.setup_fs()
            self.generatorlosses = losses.GeneratorLoss(opt, self.device)
            self.generatorlosses.dp_group = self.dp if self.dp.active else None
            if self.use_idt:
                self.idtlosses = self.generatorlosses
            self.setup_gan()
            if self.cri_gan:
                self.setup_freezeD()
            self.setup_optimizers(opt_G_nets, opt_D_nets, init_setup=True)
            self.setup_schedulers()
            self.optimizer_G.zero_grad()
            if self.cri_gan:
                self.optimizer_D.zero_grad()
            self.log_dict, self.log_dict_A, self.log_dict_B = LazyLog(), LazyLog(), LazyLog()
            self.setup_virtual_batch()
            self.setup_amp()
            self.sync_replicas()
        self.print_network(verbose=False)

    def feed_data(self, data):
        self.real_A = self._shard(data["A"]).to(self.device, non_blocking=True)
        self.real_B = self._shard(data["B"]).to(self.device, non_blocking=True)
        self.image_paths = data.get("A_path")

    def forward(self):
        self.fake_B = self.netG_A(self.real_A)
        self.rec_A = self.netG_B(self.fake_B)
        self.fake_A = self.netG_B(self.real_B)
        self.rec_B = self.netG_A(self.fake_A)

    def backward_D_A(self):
        fake_B = self.fake_B_pool.query(self.fake_B)
        self._arm_bucket_schedule([self.netD_A], passes=2)
        self.log_dict_A = self.backward_D_Basic(self.netD_A, self.real_B, fake_B, self.log_dict_A)

    def backward_D_B(self):
        fake_A = self.fake_A_pool.query(self.fake_A)
        self._arm_bucket_schedule([self.netD_B], passes=2)
        self.log_dict_B = self.backward_D_Basic(self.netD_B, self.real_A, fake_A, self.log_dict_B)

    def _acc(self, loss):
        return loss if self.accumulations == 1 else loss / self.accumulations

    def backward_G(self):
        l_g_total = 0
        if self.use_idt:
            self.idt_A = self.netG_A(self.real_B)
            self.idt_B = self.netG_B(self.real_A)
            if self.idtlosses.loss_list:
                for idt, real, log in ((self.idt_A, self.real_B, self.log_dict_A), (self.idt_B, self.real_A, self.log_dict_B)):
                    tmp = LazyLog()
                    loss_idt, tmp = self.idtlosses(idt, real, tmp, self.f_low)
                    l_g_total = l_g_total + self._acc(sum(loss_idt) * self.lambda_idt)
                    for k, v in tmp.items():
                        log["{}_idt".format(k)] = v
        if self.cri_gan:
            l_g_gan_A = self.adversarial(self.fake_B, self.real_A, netD=self.netD_A, stage="generator", fsfilter=self.f_high)
            self.log_dict_A["l_g_gan"] = self.adversarial._logged(l_g_gan_A)
            l_g_total = l_g_total + self._acc(l_g_gan_A)
            l_g_gan_B = self.adversarial(self.fake_A, self.real_B, netD=self.netD_B, stage="generator", fsfilter=self.f_high)
            self.log_dict_B["l_g_gan"] = self.adversarial._logged(l_g_gan_B)
            l_g_total = l_g_total + self._acc(l_g_gan_B)
        loss_results, self.log_dict_A = self.generatorlosses(self.rec_A, self.real_A, self.log_dict_A, self.f_low)
        l_g_total = l_g_total + self._acc(sum(loss_results))
        loss_results, self.log_dict_B = self.generatorlosses(self.rec_B, self.real_B, self.log_dict_B, self.f_low)
        l_g_total = l_g_total + self._acc(sum(loss_results))
        # each generator appears 2 (+1 with the identity term) times in the graph: buckets leave in its last pass
        self._arm_bucket_schedule([self.netG_A, self.netG_B], passes=3 if self.use_idt else 2)
        self.calc_gradients(l_g_total)
        for k, v in self.log_dict_A.items():
            self.log_dict["{}_A".format(k)] = v
        for k, v in self.log_dict_B.items():
            self.log_dict["{}_B".format(k)] = v

    @training_step
    def optimize_parameters(self, step):
        eff_step = step / self.accumulations
        self.forward()
        if self.cri_gan:
            self.requires_grad(self.netD_A, flag=False, net_type="D")
            self.requires_grad(self.netD_B, flag=False, net_type="D")
        if (self.cri_gan is not True) or (eff_step % self.D_update_ratio == 0 and eff_step > self.D_init_iters):
            self.backward_G()
            self.optimizer_step(step, self.optimizer_G, "G")
        if self.cri_gan:
            self.requires_grad(self.netD_A, True)
            self.requires_grad(self.netD_B, True)
            if isinstance(self.feature_loc, int):
                for loc in range(self.feature_loc):
                    self.requires_grad(self.netD_A, False, target_layer=loc, net_type="D")
                    self.requires_grad(self.netD_B, False, target_layer=loc, net_type="D")
            self.backward_D_A()
            self.backward_D_B()
            self.optimizer_step(step, self.optimizer_D, "D")

    def get_current_log(self, direction=None):
        src = {"A": self.log_dict_A, "B": self.log_dict_B}.get(direction, self.log_dict)
        log = src.materialize() if isinstance(src, LazyLog) else OrderedDict(src)
        self.check_engine_errors()
        return log

    def get_current_visuals(self):
        out = OrderedDict()
        for name in self.visual_names:
            out[name] = getattr(self, name).detach()[0].float().cpu()
        return out
