"""SRModel on the MI355X engine -- the drop-in for codes/models/sr_model.py.

Same constructor sequence (:22-113), `feed_data` (:115-128), `forward` (:134-160), `backward_G`
(:162-188), `backward_D` (:190-193), **`optimize_parameters`** (:195-267), `test` (:269-277) and
visuals/log accessors (:352-372), so codes/train.py drives it unchanged.  netG / netD / netF are the
HIP-engine networks (single autograd node each), the losses are HIP kernels, clip+Adam are fused over
flat buffers and gradients are exchanged with RCCL when WORLD_SIZE > 1.
"""
import logging
import os
from collections import OrderedDict

import torch

from . import losses, networks
from .base_model import BaseModel, LazyLog, training_step

logger = logging.getLogger("base")


class SRModel(BaseModel):
    def __init__(self, opt, step=0):
        super().__init__(opt)
        train_opt = opt["train"]
        self.model_names = ["G"]
        self.netG = networks.define_G(opt, step=step).to(self.device)
        if self.is_train:
            self.netG.train()
            opt_G_nets, opt_D_nets = [self.netG], []
            if train_opt["gan_weight"]:
                self.model_names.append("D")
                self.netD = networks.define_D(opt).to(self.device)
                self.netD.train()
                # D(real) and D(fake) are evaluated in the generator stage AND in the discriminator stage with unchanged
                # discriminator weights (sr_model.py:170-177,190-193): the second evaluation reuses the first (engine.HipNet.memoize)
                self.netD.memoize = os.environ.get("TNR_D_MEMO", "1") != "0"
                opt_D_nets.append(self.netD)
            self.setup_atg()
        self.load()
        self.outm = None
        if self.is_train:
            self.outm = train_opt.get("finalcap", None)
            self.setup_batchaug()
            self.setup_fs()
            self.generatorlosses = losses.GeneratorLoss(opt, self.device)
            self.generatorlosses.dp_group = self.dp if self.dp.active else None
            self.setup_gan()
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
            self.setup_cem()
            self.setup_unshuffle()
            self.setup_gradclip(opt_G_nets)
        if self.is_train:
            self.sync_replicas()
            self.calibrate_engine()
        self.print_network(verbose=False)

    def calibrate_engine(self):
        """Per-box kernel-form choices that rest on a timing, made ONCE here on scratch buffers of the training shape (this rank's
        shard of `batch_size` x crop_size / scale) and agreed on by all data-parallel ranks -- never inside a forward (ops.SWEEP_AUTO)."""
        from .. import ops
        if not hasattr(self.netG, "calibrate_dense_block_form") or not ops.SWEEP_AUTO or getattr(self, "amp", False):
            return            # (`use_amp` steps run the bf16-operand forms: no choice to make)
        ds = self.opt["datasets"]["train"]
        per = max(1, int(ds["batch_size"]) // max(1, self.dp.world_size))
        lr_size = max(8, int(ds.get("crop_size") or 128) // int(self.opt.get("scale") or 4))
        self.netG.calibrate_dense_block_form(per, lr_size, lr_size, self.dp)

    def feed_data(self, data, need_HR=True):
        self.var_L = self._shard(data["LR"]).to(self.device, non_blocking=True)
        if self.dp.world_size > 1 and hasattr(self.netG, "noise_sample0"):
            # ESRGAN+ noise: every rank draws the field of ITS samples of the global batch (what one process would draw)
            self.netG.noise_sample0 = self.dp.rank * self.var_L.shape[0]
        if need_HR:
            self.real_H = self._shard(data["HR"]).to(self.device, non_blocking=True)
            self.var_ref = self._shard(data.get("ref", data["HR"])).to(self.device, non_blocking=True)

    def feed_data_batch(self, data, need_HR=True):
        """sr_model.py:130-132: a ready LR batch (test_chop's patches)."""
        self.var_L = data

    def forward(self, data=None, CEM_net=None):
        if isinstance(data, torch.Tensor):
            return self.netG(data)
        self.fake_H = self.netG(self.var_L, outm=self.outm) if self.outm else self.netG(self.var_L)

    def backward_G(self):
        loss_results, self.log_dict = self.generatorlosses(self.fake_H, self.real_H, self.log_dict, self.f_low)
        l_g_total = sum(loss_results)
        if self.accumulations != 1:
            l_g_total = l_g_total / self.accumulations
        if self.cri_gan:
            l_g_gan = self.adversarial(self.fake_H, self.var_ref, netD=self.netD, stage="generator", fsfilter=self.f_high)
            self.log_dict["l_g_gan"] = l_g_gan.detach()
            l_g_total = l_g_total + (l_g_gan if self.accumulations == 1 else l_g_gan / self.accumulations)
        # G's gradient buckets (67 MB) CAN leave for RCCL from inside its backward, on the side stream, with the dense blocks staying one
        # launch each next to the collectives (the dispensed four-wave sweep, ops.g_buckets_leave_in_backward).  That combination --
        # tile hand-offs inside a launch while RCCL kernels that wait for PEERS hold CUs -- has never run on more than one GPU, so it is
        # opt-in (TNR_DP_OVERLAP_G=auto / 1) and the default sends G's buckets in one sweep at the optimizer step (_sync_gradients; about
        # 1 ms of exposed xGMI time per 238 ms step).  D's buckets (110 MB, the larger share) always leave inside its backward: the
        # discriminator has no one-launch dense blocks.
        from .. import ops
        if ops.g_buckets_leave_in_backward():
            self._arm_bucket_schedule([self.netG], passes=1)
        self.calc_gradients(l_g_total)

    def backward_D(self):
        self._arm_bucket_schedule([self.netD], passes=2)   # D(fake) and D(real) both accumulate: buckets leave in the 2nd
        self.log_dict = self.backward_D_Basic(self.netD, self.var_ref, self.fake_H, self.log_dict)

    @training_step
    def optimize_parameters(self, step):
        eff_step = step / self.accumulations
        if self.cri_gan:
            self.requires_grad(self.netD, flag=False, net_type="D")
        self.forward()
        if (self.cri_gan is not True) or (eff_step % self.D_update_ratio == 0 and eff_step > self.D_init_iters):
            self.backward_G()
            self.optimizer_step(step, self.optimizer_G, "G")
        if self.cri_gan:
            self.requires_grad(self.netD, flag=True)
            if isinstance(self.feature_loc, int):
                for loc in range(self.feature_loc):
                    self.requires_grad(self.netD, False, target_layer=loc, net_type="D")
            self.backward_D()
            self.optimizer_step(step, self.optimizer_D, "D")

    def test(self, CEM_net=None):
        self.netG.eval()
        with torch.no_grad():
            self.forward()
        self.netG.train()

    def test_x8(self, CEM_net=None):
        """Geometric self-ensemble (sr_model.py:277-315): the 8 flip / transpose variants of the LR batch are
        super-resolved, mapped back and averaged.  Variant i is built by the reference's doubling order
        (i bit 0: flip W, bit 1: flip H, bit 2: transpose) and undone in the order transpose, flip H, flip W;
        like the reference the result is the mean over the concatenated batch dimension ([1, C, H, W])."""
        self.netG.eval()

        def tf(v, op):
            if op == "v":
                return v.flip(3).contiguous()
            if op == "h":
                return v.flip(2).contiguous()
            return v.transpose(2, 3).contiguous()

        lr_list = [self.var_L]
        for op in ("v", "h", "t"):
            lr_list.extend([tf(t, op) for t in lr_list])
        with torch.no_grad():
            sr_list = [self.forward(data=aug) for aug in lr_list]
        for i in range(len(sr_list)):
            if i > 3:
                sr_list[i] = tf(sr_list[i], "t")
            if i % 4 > 1:
                sr_list[i] = tf(sr_list[i], "h")
            if (i % 4) % 2 == 1:
                sr_list[i] = tf(sr_list[i], "v")
        self.fake_H = torch.cat(sr_list, dim=0).mean(dim=0, keepdim=True)
        self.netG.train()

    def test_chop(self, patch_size=200, step=1.0, CEM_net=None):
        """Patch-wise inference for images that do not fit in one forward (sr_model.py:317-350): the LR image is cut
        into patch_size x patch_size windows (step = fraction of the patch between window starts, 0.5 .. 1.0), each is
        super-resolved on its own and the results are cross-faded back together.  Like the reference this handles one
        image per call (batch 1)."""
        from ..dataops.common import extract_patches_2d, recompose_tensor
        _, _, h, w = self.var_L.size()
        patch_size = min(h, w, patch_size)
        patches = extract_patches_2d(self.var_L, (patch_size, patch_size), step=[step, step], batch_first=True).squeeze(0)
        self.netG.eval()
        with torch.no_grad():
            sr = [self.forward(data=patches[i:i + 1].contiguous()) for i in range(patches.size(0))]
        self.fake_H = recompose_tensor(torch.cat(sr, 0), h, w, step=step, scale=self.opt["scale"])
        self.netG.train()

    def get_current_log(self):
        log = self.log_dict.materialize()           # the step's one host sync
        self.check_engine_errors()
        return log

    def get_current_visuals(self, need_HR=True):
        out = OrderedDict()
        out["LR"] = self.var_L.detach()[0].float().cpu()
        out["SR"] = self.fake_H.detach()[0].float().cpu()
        if need_HR:
            out["HR"] = self.real_H.detach()[0].float().cpu()
        return out

    def get_current_visuals_batch(self, need_HR=True):
        out = OrderedDict()
        out["LR"] = self.var_L.detach().float().cpu()
        out["SR"] = self.fake_H.detach().float().cpu()
        if need_HR:
            out["HR"] = self.real_H.detach().float().cpu()
        return out
