"""Glue shared by the network executors: one convolution layer bound to its packed weights, and
the autograd bridge that lets the reference-shaped training code (`loss.backward()`) drive the
hand-written forward / backward kernel schedules of a whole network as ONE autograd node.
"""
import torch

from . import hip, ops
from .flat import FlatParams


class ConvOp:
    """Forward, data-gradient and weight-gradient launches of one Conv2dHIP layer."""

    def __init__(self, mod, packer, need_dgrad=True, ups=False):
        self.mod, self.packer = mod, packer
        k, s = mod.kernel_size, mod.stride
        if k == 3 and s == 1:
            self.mode_f = ops.CONV_3x3_UP2 if ups else ops.CONV_3x3
            self.mode_d = ops.CONV_3x3
            pf, pd = ops.PACK_FWD, ops.PACK_DGRAD_3x3
        elif k == 4 and s == 2 and not ups:
            self.mode_f, self.mode_d = ops.CONV_4x4_S2, ops.DGRAD_4x4_S2
            pf, pd = ops.PACK_FWD_S2D, ops.PACK_DGRAD_S2
        else:
            raise NotImplementedError("conv k=%d s=%d is not on the SR hot path" % (k, s))
        self.i_f = packer.add(mod.weight, pf)
        self.i_d = packer.add(mod.weight, pd) if need_dgrad else None
        # deep layers that can meet a tiny spatial extent (discriminator tail): column-layout packings for the
        # im2col + split-K GEMM path (ops.conv_small)
        deep = mod.in_channels * k * k >= 2048 and not ups
        self.i_fc = packer.add(mod.weight, ops.PACK_COL_FWD) if deep else None
        self.i_dc = packer.add(mod.weight, ops.PACK_COL_DGRAD3) if (deep and need_dgrad and k == 3) else None
        self.k, self.stride = k, s
        # <= 4-channel image on either side of a 3x3 layer: taps folded into K (TNR_CONV_3x3_C4)
        img = k == 3 and s == 1 and not ups
        self.thin = img                 # <= 4 channels on the OUTPUT side of a launch: vector-ALU kernel (conv_thin.hip)
        self.i_f4 = packer.add(mod.weight, ops.PACK_C4_FWD) if (img and mod.in_channels <= 4) else None
        self.i_d4 = packer.add(mod.weight, ops.PACK_C4_DGRAD3) if (img and need_dgrad and mod.out_channels <= 4) else None

    @staticmethod
    def _image4(v):
        """The whole 4-channel pixel of an NHWC4 image view (3 channels + the zero pad), or None."""
        if ops.IMAGE_C4 and v.ctot == 4 and v.coff == 0:
            return v if v.C == 4 else ops.View(v.buf)
        return None

    @staticmethod
    def _plain(epi):
        """Only bias / alpha: what the vector-ALU thin kernel's epilogue covers."""
        return all(epi.get(n) is None for n in ("r1", "r2", "mask")) and epi.get("act", ops.ACT_NONE) == ops.ACT_NONE

    def fwd(self, x, y, **epi):
        if self.thin and ops.IMAGE_C4 and self.mod.out_channels <= 4 and y.C <= 4 and self._plain(epi):
            ops.conv_thin(x, self.mod.weight, y, bias=self.mod.bias, alpha=epi.get("alpha", 1.0))
            return
        x4 = self._image4(x) if self.i_f4 is not None else None
        if x4 is not None:
            ops.conv(x4, self.packer.get(self.i_f4), y, mode=ops.CONV_3x3_C4, bias=self.mod.bias, **epi)
            return
        if self.i_fc is not None and ops.small_gemm_ok(x, y, self.k, self.stride, epi):
            ops.conv_small(x, self.packer.get(self.i_fc), y, self.k, self.stride, bias=self.mod.bias, **epi)
            return
        ops.conv(x, self.packer.get(self.i_f), y, mode=self.mode_f, bias=self.mod.bias, **epi)

    def fwd_shuffle2(self, x, y, **epi):
        """This layer + nn.PixelShuffle(2) in one launch, y = the shuffled tensor (ops.conv_shuffle2); False: not available here."""
        if self.mode_f != ops.CONV_3x3 or self.mod.out_channels != 4 * y.C:
            return False
        return ops.conv_shuffle2(x, self.packer.get(self.i_f), y, bias=self.mod.bias, **epi)

    def fwd_stage(self, x, y, fresh_from=None, **epi):
        """Stage descriptor of this layer's forward for ops.conv_chain."""
        return dict(x=x, wp=self.packer.get(self.i_f), y=y, mode=self.mode_f, bias=self.mod.bias, fresh_from=fresh_from, **epi)

    def dgrad(self, g, gx, **epi):
        """gx = conv_transpose(g); for an UP2 layer gx lives in the up-sampled domain."""
        if self.thin and ops.IMAGE_C4 and self.mod.in_channels <= 4 and gx.C <= 4 and self._plain(epi):
            ops.conv_thin(g, self.mod.weight, gx, alpha=epi.get("alpha", 1.0), dgrad=True)
            return
        g4 = self._image4(g) if self.i_d4 is not None else None
        if g4 is not None:
            ops.conv(g4, self.packer.get(self.i_d4), gx, mode=ops.CONV_3x3_C4, **epi)
            return
        if self.i_dc is not None and ops.small_gemm_ok(g, gx, 3, 1, epi):
            ops.conv_small(g, self.packer.get(self.i_dc), gx, 3, 1, **epi)
            return
        ops.conv(g, self.packer.get(self.i_d), gx, mode=self.mode_d, **epi)

    def wgrad_item(self, x, g, alpha=1.0, cin_begin=0, with_bias=True, reflect=False):
        """Descriptor of dW[:, cin_begin : cin_begin + x.C] (+ db) += alpha * (g (x) x) for ops.wgrad_group."""
        w = self.mod.weight
        db = self.mod.bias.grad if (with_bias and self.mod.bias is not None and cin_begin == 0) else None
        return dict(x=x, g=g, dw=w.grad, db=db, cin_begin=cin_begin, alpha=alpha, beta=1.0, reflect=reflect)

    def wgrad(self, x, g, alpha=1.0, cin_begin=0, with_bias=True, reflect=False):
        if reflect:           # ReflectionPad2d(1) in front of the layer (TNR_CONV_3x3 on the matrix cores only)
            ops.wgrad_group([self.wgrad_item(x, g, alpha, cin_begin, with_bias, reflect=True)], mode=self.mode_f)
            return
        if self.thin and ops.IMAGE_C4 and cin_begin == 0 and x.N == g.N and (x.H, x.W) == (g.H, g.W):
            m = self.mod
            db = m.bias.grad if (with_bias and m.bias is not None) else None
            if m.in_channels <= 3 and x.ctot == 4 and x.coff == 0 and g.C in (16, 32, 64):      # image -> features
                ops.wgrad_thin(g, ops.View(x.buf), m.weight.grad, db, flip=False, alpha=alpha)
                return
            if m.out_channels <= 3 and g.ctot == 4 and g.coff == 0 and x.C in (16, 32, 64):     # features -> image
                ops.wgrad_thin(x, ops.View(g.buf), m.weight.grad, db, flip=True, alpha=alpha)
                return
        ops.wgrad_group([self.wgrad_item(x, g, alpha, cin_begin, with_bias)], mode=self.mode_f)


class _NetFn(torch.autograd.Function):
    """Whole-network autograd node.  Parameters are listed as inputs only so that autograd knows the
    output depends on them; their gradients are accumulated by the engine straight into the flat
    gradient buffer (param.grad views), so backward returns None for them."""

    @staticmethod
    def forward(ctx, net, x, *params):
        # (grad mode is always off inside Function.forward; needs_input_grad already accounts for it)
        need_param_grad = any(ctx.needs_input_grad[2:])
        need_graph = ctx.needs_input_grad[1] or need_param_grad
        memo = net._memo_lookup(x) if (net.memoize and net.training) else None
        if memo is not None and memo[1] is not None:
            # same input values, same parameters as an earlier forward of this step: its output and saved activations ARE this
            # forward's; only the side effects of running it again (BatchNorm running statistics) are replayed
            out, saved = memo
            net.replay_forward_side_effects(saved)
            out = out.detach()
        else:
            out, saved = net.engine_forward(x, save=need_graph or (net.memoize and net.training))
            if net.memoize and net.training:
                net._memo_store(x, out, saved)
        ctx.net, ctx.saved, ctx.need_param_grad = net, (saved if need_graph else None), need_param_grad
        return out

    @staticmethod
    def backward(ctx, gout):
        saved = ctx.saved
        if saved is None:
            raise RuntimeError("HIP engine: backward through a forward that saved no activations")
        ctx.saved = None
        sched = getattr(ctx.net, "_bucket_schedule", None)
        if sched is not None and ctx.need_param_grad:
            sched.begin_pass()                      # dp.BucketSchedule: buckets leave in the last accumulating pass
        gx = ctx.net.engine_backward(saved, gout, need_input_grad=ctx.needs_input_grad[1],
                                     need_param_grad=ctx.need_param_grad)
        return (None, gx) + (None,) * (len(ctx.needs_input_grad) - 2)


class HipNet(torch.nn.Module):
    """Base class of the engine's networks: flat parameters + the autograd bridge."""

    # Forward memoization (off by default; the SR model switches it on for its discriminator): a training-mode forward whose
    # input tensor (same storage, same version counter) and parameter version match an earlier forward is not recomputed --
    # output and saved activations are shared, backward passes only read them.  SRModel.optimize_parameters shows the
    # discriminator the real batch and the generated batch twice between two discriminator updates (generator stage, then
    # discriminator stage on `fake.detach()`): 2 of its 4 forward passes per step are such repeats.  Entries hold a reference
    # to their input (its address cannot be recycled while the entry lives) and die with the parameter version.
    memoize = False

    def _memo_key(self, x):
        fp = self.flat_params()
        return (fp.version, fp.flat._version), (x.data_ptr(), tuple(x.shape), tuple(x.stride()), x._version)

    def _memo_lookup(self, x):
        ver, key = self._memo_key(x)
        if self._memo_ver != ver:
            self._memo, self._memo_ver = {}, ver
        hit = self._memo.get(key)
        return None if hit is None else (hit[1], hit[2])

    def _memo_store(self, x, out, saved):
        ver, key = self._memo_key(x)
        if self._memo_ver != ver:
            self._memo, self._memo_ver = {}, ver
        self._memo[key] = (x.detach(), out, saved)        # (detached alias: pins the storage, not the autograd graph)

    def memo_clear(self):
        """Drop every memo entry (the models call this at the start of each training step: an entry is valid only between
        the stages of ONE step -- feeder slots are refilled by raw kernels that do not move the tensors' version counters)."""
        self._memo = {}

    def replay_forward_side_effects(self, saved):
        """Hook: what a repeated training-mode forward would change besides its outputs (BatchNorm running statistics)."""

    def _init_engine(self):
        self._memo, self._memo_ver = {}, None
        self._flat = FlatParams(self)
        self._packer = None
        self._dense_packer = None
        self._packer_owner = None
        self._packed_version = None
        self._ops = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._flat.touch())

    def flat_params(self):
        return self._flat.ensure()

    def _refresh_derived(self):
        """Hook: tensors derived from the parameters that the packer reads (zero-extended taps, ...) are rebuilt here,
        right before the weights are re-packed."""

    def _prepare(self, x):
        hip.require_device(x)
        if x.dtype != torch.float32:
            raise hip.HipEngineError("the HIP engine computes in fp32; got %s" % x.dtype)
        fp = self.flat_params()
        if fp.flat.device != x.device:
            raise hip.HipEngineError("network is on %s but input on %s" % (fp.flat.device, x.device))
        if self._packer is None or self._packer.device != x.device or self._packer_owner is not fp.flat:
            self._packer = ops.WeightPacker(x.device)
            self._dense_packer = ops.DensePacker(x.device)
            self._packer_owner = fp.flat
            self._packed_version = None
            self._build_ops(self._packer)
        # parameters only change through FusedAdam.step / load_state_dict / re-flattening, all of which bump
        # fp.version: the D (4 forwards per step) and the frozen VGG are not re-packed needlessly
        # (fp.flat._version also moves on any in-place torch write through a parameter view)
        key = (fp.version, fp.flat._version)
        if self._packed_version != key:
            self._refresh_derived()
            self._packer.run()
            self._dense_packer.run()
            self._packed_version = key

    def forward(self, x, **kw):
        x = x.contiguous()
        self._prepare(x)
        params = list(self.parameters())
        return _NetFn.apply(self, x, *params)
