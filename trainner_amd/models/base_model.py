"""BaseModel of the MI355X engine: the reference's model-object protocol (codes/models/base_model.py).

Kept verbatim in behaviour: device pick (:76-81), save/load of networks as legacy-pickle CPU
state_dicts with `latest_/previous_` rotation (:353-443), training-state save/resume (:454-500), LR /
scheduler plumbing (:209-317), `requires_grad` (:325-351), the `setup_*` feature switches (:641-787) and
`calc_gradients` / `optimizer_step` / `backward_D_Basic` / `apply_gradclip` (:805-922).
Re-designed: gradient clipping and Adam act on the flat buffers (2 + 1 launches per network), the
data-parallel gradient exchange happens inside `optimizer_step`, AMP/SWA/CEM/ATG/batch-aug switches
raise if enabled (they are off in the ESRGAN recipe and outside the hot path, SURVEY.md 2 #18,#24).
"""
import logging
import os
from collections import Counter, OrderedDict
from contextlib import contextmanager, nullcontext
from shutil import copyfile

import torch

from .. import dp as dpmod
from .. import hip, ops
from . import optimizers, schedulers
from .losses import Adversarial
from .networks import model_val

logger = logging.getLogger("base")


def training_step(fn):
    """Decorator of the models' optimize_parameters: the step runs inside the AMP region (`use_amp`: bf16 matrix-core operands
    for forward, losses and backward; validation stays fp32) and starts with empty forward memos -- a memo entry is only ever
    valid between the generator stage and the discriminator stage of ONE step (engine.HipNet.memoize): buffers refilled by raw
    kernels (DeviceFeeder slots) keep their torch version counter, so entries must not survive into a later batch."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, step, *a, **kw):
        for name in self.model_names:
            net = getattr(self, "net" + name, None)
            if net is not None and hasattr(net, "memo_clear"):
                net.memo_clear()
        with self.amp_region():
            return fn(self, step, *a, **kw)
    return wrapped


class LazyLog(OrderedDict):
    """log_dict whose values may be 0-dim device tensors; they are fetched (one host sync per read
    of the dict, not eight per step like the reference's `.item()` calls: losses.py:862,
    sr_model.py:177, losses.py:515-520) when the training script asks for the log."""

    def materialize(self):
        out = OrderedDict()
        for k, v in self.items():
            out[k] = float(v) if torch.is_tensor(v) else v
        return out


class BaseModel:
    def __init__(self, opt):
        self.opt = opt
        if opt["gpu_ids"]:
            hip.require_device()
            local = int(os.environ.get("TNR_DP_DEVICE", os.environ.get("LOCAL_RANK", "0")))      # (TNR_DP_DEVICE: see dp.init_from_env)
            self.device = hip.engine_device(local)
            if self.device.type == "cuda":
                torch.cuda.set_device(self.device)
        else:
            # the reference falls back to CPU here (base_model.py:80-81); this engine has no CPU path
            raise hip.HipEngineError("gpu_ids is empty: the trainner_amd engine runs only on MI355X "
                                     "(use the reference itself, or oracle/, for CPU)")
        self.is_train = opt["is_train"]
        self.model_names = []
        self.schedulers = []
        self.optimizers = []
        self.swa = None
        self.swa_start_iter = None
        self.metric = 0
        self.batchaugment = None
        self.upsample = False
        self.unshuffle = None
        self.grad_clip = None
        self.grad_history = []
        self.dp = dpmod.init_from_env()

    # ------------------------------------------------------------------ protocol stubs
    def _shard(self, t):
        """Reference semantics of `batch_size` (options/README.md:31: the GLOBAL batch, split over gpu_ids by
        nn.DataParallel's scatter): a fed tensor that carries the global batch is cut to this rank's contiguous
        shard; a tensor that already holds batch_size / world samples (a per-rank loader) is taken as is."""
        world = self.dp.world_size
        if world == 1 or not self.is_train:
            return t
        gb = self.opt["datasets"]["train"]["batch_size"]
        if gb % world:
            raise ValueError("batch_size %d is not divisible by the %d data-parallel ranks" % (gb, world))
        per = gb // world
        if t.shape[0] == gb:
            return t[self.dp.rank * per:(self.dp.rank + 1) * per]
        if t.shape[0] == per:
            return t
        raise ValueError("fed batch of %d samples is neither the global batch (%d) nor this rank's shard (%d)"
                         % (t.shape[0], gb, per))

    def feed_data(self, data):
        pass

    def optimize_parameters(self, step):
        pass

    def get_current_visuals(self):
        pass

    def get_current_losses(self):
        pass

    def print_network(self, verbose=False):
        for name in self.model_names:
            net = getattr(self, "net" + name)
            s, n = self.get_network_description(net)
            logger.info("Network %s structure: %s, with parameters: %s", name, net.__class__.__name__, "{:,d}".format(n))
            if verbose:
                logger.info(s)

    def get_network_description(self, network):
        return str(network), sum(p.numel() for p in network.parameters())

    def check_engine_errors(self):
        """Host-side read of the fault latch (ops.fault_word) at points that synchronise anyway (log read-out, checkpoint).  A tile
        hand-off wait of a one-launch dense block that timed out computed with unpublished neighbour tiles: the latch is sticky and
        every Adam launch runs behind it ON THE DEVICE (tnr_adam_step_guarded), so no optimiser step is applied from the faulted
        launch on -- the weights here are those of the last healthy step -- and the run stops at this check.  Data-parallel ranks
        share the latch (MAX-reduced with every gradient exchange, _sync_gradients): all of them stop at the same step.  Only the
        weights and Adam's moments are protected; BatchNorm running statistics and the step counters of the faulted step have moved."""
        if ops.chain_error_flag():
            raise hip.HipEngineError("a tile hand-off wait inside a one-launch dense block timed out (another tenant on the GPU kept "
                                     "the workgroups from being resident?): no optimiser step has been applied since; restart with "
                                     "TNR_CONV_CHAIN=0 to run one launch per layer")

    # ------------------------------------------------------------------ checkpoints
    def save(self, iter_step, latest=None, loader=None):
        self.check_engine_errors()
        if self.dp.rank != 0:
            return
        for name in self.model_names:
            self.save_network(getattr(self, "net" + name), name, iter_step, latest)

    def load(self):
        for name in self.model_names:
            load_path = self.opt["path"]["pretrain_model_{}".format(name)]
            if load_path is None:
                continue
            logger.info("Loading pretrained model for %s [%s]", name, load_path)
            # strict flag of this net's own option block, else of its family's (G_A -> network_G; base_model.py:201-206)
            strict = True
            for key in ("network_{}".format(name), "network_{}".format("D" if "D" in name else "G")):
                if self.opt.get(key):
                    strict = self.opt[key].get("strict", None)
                    break
            self.load_network(load_path, getattr(self, "net" + name), strict, model_type=name)

    def save_network(self, network, network_label, iter_step, latest=False):
        fname = "latest_{}.pth".format(network_label) if latest else "{}_{}.pth".format(iter_step, network_label)
        save_path = os.path.join(self.opt["path"]["models"], fname)
        os.makedirs(os.path.dirname(save_path), exist_ok=True)
        if os.path.exists(save_path):
            copyfile(save_path, os.path.join(self.opt["path"]["models"], "previous_{}.pth".format(network_label)))
        state = OrderedDict((k, v.detach().cpu().clone()) for k, v in network.state_dict().items())
        torch.save(state, save_path, _use_new_zipfile_serialization=False)

    def load_network(self, load_path, network, strict=True, submodule=None, model_type=None, param_key=None):
        load_net = torch.load(load_path, map_location="cpu", weights_only=False)
        if "state_dict" in load_net:
            load_net = load_net["state_dict"]
        if param_key is not None:
            load_net = load_net[param_key]
        if model_type:                     # e.g. new-arch ESRGAN keys -> this package's layout
            load_net = model_val(opt_net=self.opt, state_dict=load_net, model_type=model_type)
        own = network.state_dict()
        if len(load_net) == len(own):      # same layout, mismatching shapes keep the initialised tensor
            load_net = OrderedDict((k, v if v.size() == own[k].size() else own[k])
                                   for k, v in zip(own.keys(), load_net.values()))
        network.load_state_dict(load_net, strict=bool(strict))

    def save_training_state(self, epoch, iter_step, latest=False):
        if self.dp.rank != 0:
            return
        state = {"epoch": epoch, "iter": iter_step, "schedulers": [s.state_dict() for s in self.schedulers],
                 "optimizers": [o.state_dict() for o in self.optimizers]}
        # engine-only entry (the reference's resume reads the four keys above and ignores the rest): the position of every network's
        # counter-based ESRGAN+ noise stream, so that a resumed run continues the stream instead of replaying it from forward 0
        noise = {name: [getattr(getattr(self, "net" + name), "_noise_calls"), getattr(getattr(self, "net" + name), "noise_seed")]
                 for name in self.model_names if hasattr(getattr(self, "net" + name), "_noise_calls")}
        if noise:
            state["tnr_engine"] = {"noise_streams": noise}
        fname = "latest.state" if latest else "{}.state".format(iter_step)
        save_path = os.path.join(self.opt["path"]["training_state"], fname)
        os.makedirs(os.path.dirname(save_path), exist_ok=True)
        if os.path.exists(save_path):
            copyfile(save_path, os.path.join(self.opt["path"]["training_state"], "previous.state"))
        torch.save(state, save_path)

    def resume_training(self, resume_state):
        ro, rs = resume_state["optimizers"], resume_state["schedulers"]
        assert len(ro) == len(self.optimizers), "Wrong length of optimizers"
        assert len(rs) == len(self.schedulers), "Wrong length of schedulers"
        for o, st in zip(self.optimizers, ro):
            o.load_state_dict(st)
        for s, st in zip(self.schedulers, rs):
            if hasattr(s, "milestones") and isinstance(s.milestones, Counter) and isinstance(st.get("milestones"), list):
                st["milestones"] = Counter(st["milestones"])
            s.load_state_dict(st)
        for name, (calls, seed) in (resume_state.get("tnr_engine") or {}).get("noise_streams", {}).items():
            net = getattr(self, "net" + name, None)
            if net is not None and hasattr(net, "_noise_calls"):
                net._noise_calls = int(calls)
                if seed is not None and net.noise_seed is None:
                    net.noise_seed = int(seed)
        self.sync_replicas()

    def update_schedulers(self, train_opt):
        for s in self.schedulers:
            if train_opt["lr_gamma"] is not None and getattr(s, "gamma", None) not in (None, train_opt["lr_gamma"]):
                s.gamma = train_opt["lr_gamma"]
            if train_opt["lr_scheme"] == "MultiStepLR" and train_opt["lr_steps"] is not None:
                steps = Counter(train_opt["lr_steps"])
                if s.milestones != steps:
                    s.milestones = steps

    # ------------------------------------------------------------------ learning rate
    def _set_lr(self, lr_groups_l):
        for optimizer, lr_groups in zip(self.optimizers, lr_groups_l):
            for group, lr in zip(optimizer.param_groups, lr_groups):
                group["lr"] = lr

    def _get_init_lr(self):
        return [[g["initial_lr"] for g in o.param_groups] for o in self.optimizers]

    def update_learning_rate(self, current_step=None, warmup_iter=-1):
        for s in self.schedulers:
            s.step()
        if current_step is not None and current_step < warmup_iter:
            self._set_lr([[v / warmup_iter * current_step for v in grp] for grp in self._get_init_lr()])
        self.optGstep = False
        if self.cri_gan:
            self.optDstep = False

    def get_current_learning_rate(self, current_step=None):
        return self.schedulers[0].get_last_lr()[0]

    def requires_grad(self, model, flag=True, target_layer=None, net_type=None):
        for name, param in model.named_parameters():
            if target_layer is None:
                param.requires_grad = flag
            elif net_type == "D":
                # vgg-like D: features.<i>. ; PatchGAN: model.<i>. (base_model.py:343-351)
                prefix = "features." if "features." in name else ("model." if "model." in name else None)
                if prefix and "{}{}.".format(prefix, target_layer) in name:
                    param.requires_grad = flag

    # ------------------------------------------------------------------ feature switches
    def _reject(self, enabled, what):
        if enabled:
            raise NotImplementedError("{} is off in the ESRGAN recipe and not implemented by the HIP engine".format(what))

    def setup_atg(self):
        self.atg = False
        self._reject(self.opt.get("use_atg"), "AdaTarget (use_atg)")

    def setup_batchaug(self):
        self.mixup = None
        self._reject(self.opt["train"].get("mixup"), "batch augmentation (mixup)")

    def setup_fs(self):
        self.f_low = self.f_high = None
        self._reject(self.opt["train"].get("fs"), "frequency separation (fs)")

    def setup_gan(self, conditional=False):
        train_opt = self.opt["train"]
        if train_opt["gan_type"] and train_opt["gan_weight"]:
            self.cri_gan = True
            self.adversarial = Adversarial(train_opt=train_opt, device=self.device, diffaug=train_opt.get("diffaug"),
                                           conditional=conditional)
            self.adversarial.dp_group = self.dp if self.dp.active else None
            self.D_update_ratio = train_opt.get("D_update_ratio", 1) or 1
            self.D_init_iters = train_opt.get("D_init_iters", 0) or 0
            logger.info("GAN enabled")
        else:
            self.cri_gan = False

    def setup_freezeD(self):
        self.feature_loc = None
        loc = self.opt["train"].get("freeze_loc")
        disc = self.opt["network_D"].get("type", "") if loc else ""
        if loc and "discriminator_vgg" in disc:
            self.feature_loc = (loc * 3) - 2
            logger.info("FreezeD enabled")
        elif loc and "patchgan" in disc:
            self.feature_loc = (loc * 3) - 1
            logger.info("FreezeD enabled")

    def setup_optimizers(self, opt_G_nets, opt_D_nets, init_setup=False):
        if init_setup:
            self.optGstep, self.optDstep = False, True
        train_opt = self.opt["train"]
        self.optimizer_G = optimizers.config_optimizer(train_opt, "G", opt_G_nets)
        self.optimizers.append(self.optimizer_G)
        self._opt_nets = {"G": list(opt_G_nets)}
        if self.cri_gan:
            self.optimizer_D = optimizers.config_optimizer(train_opt, "D", opt_D_nets)
            self.optimizers.append(self.optimizer_D)
            self._opt_nets["D"] = list(opt_D_nets)
            self.optDstep = False

    def setup_schedulers(self):
        self.schedulers = schedulers.get_schedulers(optimizers=self.optimizers, train_opt=self.opt["train"])

    def setup_swa(self):
        self.swa = None
        self._reject(self.opt.get("use_swa"), "SWA (use_swa)")

    def setup_virtual_batch(self):
        ds = self.opt["datasets"]["train"]
        batch_size = ds["batch_size"]
        vb = ds.get("virtual_batch_size", None)
        self.virtual_batch = vb if (vb and vb > batch_size) else batch_size
        self.accumulations = self.virtual_batch // batch_size

    def setup_amp(self):
        """`use_amp: true` (the reference's ESRGAN recipe ships it: options/sr/train_sr.yml:6; base_model.py:736-744 wraps
        the forward / loss code in torch.cuda.amp.autocast + GradScaler, i.e. fp16 convolutions with fp32 master weights).
        The engine's policy: every matrix-core launch (convolutions forward / data-gradient / weight-gradient of G, D and
        the VGG feature net) rounds its operands to bf16 on their way into v_mfma_f32_32x32x16_bf16 and accumulates in
        fp32; activations, master weights, BatchNorm, the losses, clip + Adam stay fp32.  bf16 has fp32's exponent
        range, so there is no loss scaling (no GradScaler state: the reference never saves it either, base_model.py:466).
        Like the reference's autocast region the mode covers the TRAINING step only (`amp_region()` around forward / losses /
        backward in optimize_parameters): test() / validation forwards run in the fp32 arithmetic (base_model.py:736-744,
        sr_model.py:269-277)."""
        self.amp = bool(self.opt.get("use_amp"))
        self.cast = self.amp_region if self.amp else nullcontext
        self.amp_scaler = None
        if self.amp:
            logger.info("AMP enabled: bf16 matrix-core operands, fp32 accumulation and master weights.")

    @contextmanager
    def amp_region(self):
        """ops.MMA = bf16 operands inside, restored on exit (also when the step raises)."""
        if not getattr(self, "amp", False):
            yield
            return
        prev, ops.MMA = ops.MMA, hip.MMA_BF16
        try:
            yield
        finally:
            ops.MMA = prev

    def setup_cem(self):
        self.CEM = None
        self._reject(self.opt.get("use_cem"), "CEM (use_cem)")

    def setup_unshuffle(self):
        self._reject(self.opt.get("use_unshuffle"), "pixel unshuffle wrapper")

    def setup_gradclip(self, clip_nets):
        train_opt = self.opt["train"]
        grad_clip = train_opt.get("grad_clip")
        if grad_clip:
            if grad_clip.lower() != "norm":
                raise NotImplementedError("grad_clip [{}] is not implemented by the HIP engine".format(grad_clip))
            self.grad_clip = "norm"
            self.grad_clip_value = train_opt.get("grad_clip_value", 0.1)      # a number, or 'auto' (get_auto_norm)
            self.clip_nets = clip_nets
            logger.info("norm gradient clip enabled. Clip value: %s.", self.grad_clip_value)

    # ------------------------------------------------------------------ step pieces
    def calc_gradients(self, loss):
        loss.backward()

    def _arm_bucket_schedule(self, nets, passes=1):
        """Let the next backward pass(es) of `nets` hand finished gradient buckets to RCCL while backward is still
        running (only when this backward completes the virtual batch: partial sums must not be reduced).
        `passes`: autograd nodes of the net in the loss (D step: D(fake) and D(real) = 2)."""
        if self.dp.active and self.accumulations == 1:
            for net in nets:
                net._bucket_schedule = dpmod.BucketSchedule(self.dp, net.flat_params(), passes=passes)

    def sync_replicas(self):
        """Data-parallel start state: every rank takes rank 0's parameters, BatchNorm buffers and Adam moments
        (each rank ran its own init RNG / load).  Called at the end of construction and after resume_training."""
        if not self.dp.active:
            return
        tensors = []
        for name in self.model_names:
            net = getattr(self, "net" + name)
            tensors.append(net.flat_params().flat)
            tensors += [b for b in net.buffers() if b.is_floating_point() or b.dtype == torch.int64]
        for o in self.optimizers:
            for mv in getattr(o, "_moments", {}).values():
                tensors += list(mv)
        # ESRGAN+ noise: ONE stream for all replicas (each rank draws the field of its own samples of the global batch): rank 0's
        # seed -- torch.initial_seed() differs per rank when manual_seed is unset -- and rank 0's position in the stream
        noisy = [getattr(self, "net" + n) for n in self.model_names if hasattr(getattr(self, "net" + n), "_noise_calls")]
        dev = tensors[0].device
        meta = torch.tensor([v for net in noisy for v in ((torch.initial_seed() if net.noise_seed is None else int(net.noise_seed)) & ((1 << 62) - 1),
                                                              net._noise_calls)], dtype=torch.int64, device=dev) if noisy else None
        if meta is not None:
            tensors.append(meta)
        self.dp.broadcast_from_rank0(tensors)
        if meta is not None:
            vals = meta.cpu().tolist()
            for i, net in enumerate(noisy):
                net.noise_seed, net._noise_calls = int(vals[2 * i]), int(vals[2 * i + 1])
        for name in self.model_names:
            getattr(self, "net" + name).flat_params().touch()

    def _zero_frozen_grads(self, opt_flag):
        """Parameters with requires_grad False at step time (FreezeD: base_model.py:641-655, sr_model.py:249-253) get
        no gradient in the reference and Adam skips them.  The engine's flat Adam launch covers the whole buffer, so
        their gradient slices are cleared: with zero gradient history Adam's update is exactly 0 (weight decay on a
        frozen tensor would not be, hence the check)."""
        for net in self._opt_nets[opt_flag]:
            frozen = [p for p in net.parameters() if not p.requires_grad]
            if not frozen:
                continue
            opt = self.optimizer_G if opt_flag == "G" else self.optimizer_D
            if any(g["weight_decay"] for g in opt.param_groups):
                raise NotImplementedError("frozen parameters with weight decay are not implemented by the HIP engine")
            for p in frozen:
                ops.fill(p.grad, 0.0)

    def _sync_gradients(self, opt_flag):
        """Data-parallel exchange of the gradients the backward pass just produced."""
        if not self.dp.active:
            return
        timed = getattr(self, "comm_events", None) is not None and self.device.type == "cuda"
        if timed:                                   # bench.py: the EXPOSED part of the exchange = how long the compute stream sits
            e0 = torch.cuda.Event(enable_timing=True)      # between the last backward kernel and the first optimiser kernel
            e0.record(torch.cuda.current_stream(self.device))
        # the fault latch is agreed on with the gradients: a rank whose dense-block launch faulted would skip its Adam step
        # alone, the replicas would diverge and its peers would block in the next all-reduce while it raises.  MAX over the ranks on
        # the compute stream, in front of the guarded Adam launches: from the faulted step on NO rank applies an update and every
        # rank's own check_engine_errors() raises.  (What the latch protects is the WEIGHTS and Adam's moments; BatchNorm running
        # statistics, schedulers and FusedAdam's host-side step counter of the faulted step have already advanced -- the run stops
        # at the next host check, it is not resumable from memory.)
        if ops._FAULT is not None and ops._FAULT.device.type == self.device.type:
            self.dp.all_reduce_max(ops._FAULT)
        for net in self._opt_nets[opt_flag]:
            holder = net.flat_params()
            sched = getattr(net, "_bucket_schedule", None)
            if sched is not None:
                sched.flush()
                net._bucket_schedule = None
            else:
                self.dp.reduce_flat(holder.grad)
        self.dp.wait(self.device)
        if timed:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(torch.cuda.current_stream(self.device))
            self.comm_events.append((opt_flag, e0, e1))

    def optimizer_step(self, step, optimizer, opt_flag):
        """base_model.py:815-850: step only when the virtual batch is complete; G gets the clip."""
        if step % self.accumulations != 0:
            return
        self._sync_gradients(opt_flag)
        self._zero_frozen_grads(opt_flag)          # after the exchange: buckets of the frozen layers may still be in flight before it
        if opt_flag == "G":
            self.apply_gradclip()
        optimizer.step()
        optimizer.zero_grad()
        if opt_flag == "G":
            self.optGstep = True
        elif opt_flag == "D":
            self.optDstep = True

    def backward_D_Basic(self, netD, real=None, fake=None, log_dict=None, condition=None):
        if log_dict is None:
            log_dict = LazyLog()
        l_d_total, gan_logs = self.adversarial(fake, real, condition, netD=netD, stage="discriminator", fsfilter=self.f_high)
        for k, v in gan_logs.items():
            log_dict[k] = v
        if self.accumulations != 1:
            l_d_total = l_d_total / self.accumulations
        self.calc_gradients(l_d_total)
        return log_dict

    def apply_gradclip(self):
        """clip_grad_norm_ over each clip net (base_model.py:911-922) = sum-of-squares reduction + one
        scaling launch over the flat gradient buffer."""
        if self.grad_clip is None:
            return
        sums = []
        for net in self.clip_nets:
            holder = net.flat_params()
            ss = torch.empty(1, dtype=torch.float64, device=holder.grad.device)
            ops.sumsq(holder.grad, ss)
            sums.append((holder, ss))
        value = self.get_auto_norm(sums=sums) if self.grad_clip_value == "auto" else self.grad_clip_value
        for holder, ss in sums:
            ops.clip_by_norm(holder.grad, ss, float(value))

    def calc_gradnorm(self, net):
        """L2 norm of a network's gradient (base_model.py:885-894), from the flat gradient buffer."""
        holder = net.flat_params()
        ss = torch.empty(1, dtype=torch.float64, device=holder.grad.device)
        ops.sumsq(holder.grad, ss)
        return float(ss.item()) ** 0.5

    def get_auto_norm(self, clip_percentile=10, sums=None):
        """`grad_clip_value: auto` (base_model.py:896-909): the clip norm is the 10th percentile of the history of per-step
        gradient norms (mean over the clipped networks).  Like the reference this reads the norm back every step."""
        norms = [float(ss.item()) ** 0.5 for _, ss in sums] if sums is not None else [self.calc_gradnorm(n) for n in self.clip_nets]
        self.grad_history.append(sum(norms) / len(norms))
        return torch.quantile(torch.FloatTensor(self.grad_history), clip_percentile / 100)
