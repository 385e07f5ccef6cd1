"""ResnetGenerator (CycleGAN / Pix2Pix generator) on the MI355X engine.

Constructor, `state_dict` keys and arithmetic follow codes/models/modules/architectures/ResNet_arch.py:11-149:
ReflectionPad(3) + conv7x7(in -> ngf) + norm + ReLU; two conv3x3 stride 2 + norm + ReLU; n_blocks x [x + (ReflectionPad(1) +
conv3x3 + norm + ReLU + ReflectionPad(1) + conv3x3 + norm)]; two ConvTranspose2d(k3, s2, p1, output_padding 1) + norm + ReLU;
ReflectionPad(3) + conv7x7(ngf -> out) + Tanh.  norm = InstanceNorm2d (no affine, no running statistics; convolutions
then carry a bias) or BatchNorm2d (train mode).  `self.model` is an nn.Sequential with the reference's child indices.

Kernel mapping (all fp32 NHWC):
  * residual blocks (>= 90 % of the FLOP): forward and weight gradient run the 3x3 MFMA kernels on the image grid with the
    reflection done by the stager (tnr_conv_desc.pad_mode 1: row -1 is read as row 1, row H as row H - 2) -- no padded tensor;
    the data-gradient is taken with respect to the padded input (zero-embedded gradient -> 3x3 data-gradient kernel on the
    (H + 2) x (W + 2) grid) and folded back by the adjoint of the reflection (tnr_unpad2d);
  * stride-2 3x3 convolutions and the transposed convolutions: the 4x4 stride-2 MFMA kernels with the 3x3 taps zero-extended
    to 4x4 (3x3 s2 p1 == 4x4 s2 p1 with a zero fourth row / column; ConvTranspose2d(k3,s2,p1,op1) == the data-gradient of
    that convolution, its input gradient == the forward, its weight gradient == the weight gradient with roles swapped);
  * the two 7x7 reflection-padded image-side convolutions (_K7Image): ONE launch per pass -- the taps-in-K MFMA mode
    (TNR_CONV_7x7_C4: 49 taps x 4 image channels = K 196 -> 208) towards the feature side, the vector-ALU kernels tnr_conv_thin7 /
    tnr_wgrad_thin7 towards the image side, reflection applied where the operand is gathered (no padded copy of a wide tensor);
    TNR_K7_FUSED=0 keeps the earlier form, nine 3x3 blocks of taps over shifted copies of the 4-channel operand (A/B switch);
    the generic kernels of csrc/gconv.hip are the fallback for channel counts those kernels do not take;
  * normalisation: the BatchNorm-train kernels (InstanceNorm: one statistics group per image in the same launches:
    tnr_instnorm_*), ReLU fused; Tanh as one pass.
"""
import os

import torch
import torch.nn as nn

from .... import ops
from ....engine import ConvOp, HipNet
from ....ops import View, new_act
from . import block as B


class ConvTranspose2dHIP(nn.Module):
    """nn.ConvTranspose2d's tensors: weight [in, out, k, k] (class name contains 'Conv' for init_weights)."""

    def __init__(self, in_nc, out_nc, kernel_size=3, stride=2, bias=True):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size, self.stride = in_nc, out_nc, kernel_size, stride
        self.weight = nn.Parameter(torch.empty(in_nc, out_nc, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_nc)) if bias else None
        # nn.ConvTranspose2d's default init, draw for draw (the same seed then gives the reference's initial weights: init_weights
        # runs after construction and its draws follow these in the generator's stream)
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        if bias:
            bound = 1.0 / (out_nc * kernel_size * kernel_size) ** 0.5        # fan_in of the [in, out, k, k] layout = size(1) * k * k
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        raise RuntimeError("ConvTranspose2dHIP is executed by its owning network's HIP engine")


class ResnetBlock(nn.Module):
    """x + conv_block(x) (ResNet_arch.py:96-149); child indices of `conv_block` as in the reference (reflect padding, no dropout)."""

    def __init__(self, dim, norm, use_bias):
        super().__init__()
        self.conv_block = nn.Sequential(B.Marker("reflectpad1"), B.Conv2dHIP(dim, dim, 3, 1, bias=use_bias), _norm_module(norm, dim),
                                        B.Marker("act:relu"), B.Marker("reflectpad1"), B.Conv2dHIP(dim, dim, 3, 1, bias=use_bias),
                                        _norm_module(norm, dim))


def _norm_module(norm, nc):
    return B.BatchNorm2dHIP(nc, affine=True) if norm == "batch" else B.Marker("instancenorm")


class _Padded4x4:
    """A 3x3 stride-2 (transposed) convolution on the 4x4 stride-2 MFMA kernels: a zero-extended [O, I, 4, 4] copy of the
    3x3 weight is what gets packed; its gradient comes back through the same copy."""

    def __init__(self, weight, packer, transposed):
        o, i = weight.shape[0], weight.shape[1]
        self.weight, self.transposed = weight, transposed
        self.w4 = torch.zeros((o, i, 4, 4), dtype=torch.float32, device=weight.device)
        self.dw4 = torch.zeros_like(self.w4)
        self.packer = packer
        self.i_f = packer.add(self.w4, ops.PACK_FWD_S2D)
        self.i_d = packer.add(self.w4, ops.PACK_DGRAD_S2)

    def refresh(self):
        self.w4[:, :, :3, :3].copy_(self.weight.detach())          # (the fourth row / column stays zero)

    def conv(self, x, y, **epi):            # large -> small: the 3x3 s2 convolution (or the transposed layer's input gradient)
        ops.conv(x, self.packer.get(self.i_f), y, mode=ops.CONV_4x4_S2, **epi)

    def conv_t(self, x, y, **epi):          # small -> large: its data-gradient (or the transposed layer's forward)
        ops.conv(x, self.packer.get(self.i_d), y, mode=ops.DGRAD_4x4_S2, **epi)

    def wgrad(self, x_large, g_small):
        ops.wgrad(x_large, g_small, self.dw4, None, mode=ops.CONV_4x4_S2, beta=0.0)
        self.weight.grad.add_(self.dw4[:, :, :3, :3])


K7_FUSED = os.environ.get("TNR_K7_FUSED", "1") != "0"      # the 7x7 image-side layers as one launch per pass (0: nine 3x3 blocks; A/B switch)
_K7_BLOCKS = tuple((by, bx) for by in (0, 3, 4) for bx in (0, 3, 4))     # first tap of each 3x3 block; a block at 4 owns tap 6 only


def _k7_first(b):
    return 2 if b == 4 else 0


class _K7Image:
    """A 7x7 stride-1 convolution over a ReflectionPad2d(3) input with a <= 4-channel image on one side (ResNet_arch.py:52-55:
    image -> ngf, :86-88: ngf -> image), as the sum of nine 3x3 blocks of its taps: with x_p the reflection-padded input,
    y[p] = sum_k w[k] x_p[p + k] = sum_blocks sum_t W_b[t] x_p[p + b + t].  Every block runs on the 3x3 kernels that exist for
    image layers (TNR_CONV_3x3_C4 on the matrix cores towards the feature side, tnr_conv_thin / tnr_wgrad_thin on the vector
    ALUs towards the image side); the offset b is applied to the 4-channel operand (tnr_window2d: the shifted image, the
    embedded image-side gradient, or the 4-channel block results), never to the wide one.  Data-gradients are taken with
    respect to the padded input and folded back (tnr_unpad2d, adjoint of the reflection)."""

    def __init__(self, conv, packer, side):
        self.mod, self.packer, self.side = conv, packer, side        # side: "in" (image -> features) | "out" (features -> image)
        w = conv.weight
        O, I = w.shape[0], w.shape[1]
        self.fused = K7_FUSED
        if self.fused:      # the layer's own 7x7 weight in the taps-in-K packing (repacked with the other weights after every step)
            self.i_k7 = packer.add(w, ops.PACK_C4_FWD if side == "in" else ops.PACK_C4_DGRAD3)
            return
        self.ws = torch.zeros((9, O, I, 3, 3), dtype=torch.float32, device=w.device)
        self.dws = torch.zeros_like(self.ws)
        self.db = torch.zeros(O, dtype=torch.float32, device=w.device)
        kind = ops.PACK_C4_FWD if side == "in" else ops.PACK_C4_DGRAD3
        self.i_mma = [packer.add(self.ws[b], kind) for b in range(9)]

    def refresh(self):
        if self.fused:
            return
        w = self.mod.weight.detach()
        self.ws.zero_()
        for b, (by, bx) in enumerate(_K7_BLOCKS):
            ty, tx = _k7_first(by), _k7_first(bx)
            self.ws[b, :, :, ty:, tx:].copy_(w[:, :, by + ty:by + 3, bx + tx:bx + 3])

    def _gather_grad(self, bias_grad):
        gw = self.mod.weight.grad
        for b, (by, bx) in enumerate(_K7_BLOCKS):
            ty, tx = _k7_first(by), _k7_first(bx)
            gw[:, :, by + ty:by + 3, bx + tx:bx + 3].add_(self.dws[b, :, :, ty:, tx:])
        if bias_grad and self.mod.bias is not None:
            self.mod.bias.grad.add_(self.db)

    # ---- image -> features (first layer): x4 [N,H,W,4] -> z [N,H,W,O]
    def fwd_in(self, x4, z):
        N, H, W, dev = x4.N, x4.H, x4.W, x4.buf.device
        if self.fused:
            ops.conv(x4, self.packer.get(self.i_k7), z, mode=ops.CONV_7x7_C4, bias=self.mod.bias, reflect=True)
            return x4                                                 # (the image itself: the weight gradient pads it again, 4 channels)
        xp = View(new_act(N, H + 6, W + 6, 4, dev))
        ops.pad2d(x4, xp, 3, True)
        acc = View(new_act(N, H + 2, W + 2, z.C, dev))
        shifted = []
        for b, (by, bx) in enumerate(_K7_BLOCKS):
            sh = View(new_act(N, H + 2, W + 2, 4, dev))
            ops.window2d(xp, sh, by, bx)                              # sh[u] = x_p[u + b]: block b reads sh[u + t - 1]
            epi = dict(bias=self.mod.bias) if b == 0 else dict(r1=acc, beta1=1.0)
            ops.conv(sh, self.packer.get(self.i_mma[b]), acc, mode=ops.CONV_3x3_C4, **epi)
            shifted.append(sh)
        ops.unpad2d(acc, z, 1, False)
        return shifted

    def bwd_in(self, shifted, gz, want_w, gx4):
        """gz: gradient of this layer's output; gx4 (or None): [N,H,W,4] gradient of the image."""
        N, H, W, dev = gz.N, gz.H, gz.W, gz.buf.device
        if self.fused:
            if want_w:
                xp = View(new_act(N, H + 6, W + 6, 4, dev))
                ops.pad2d(shifted, xp, 3, True)                       # (shifted: the NHWC4 image fwd_in returned)
                ops.wgrad_thin7(gz, xp, self.mod.weight.grad, self.mod.bias.grad if self.mod.bias is not None else None, flip=False)
            if gx4 is None:
                return
            gxp = View(torch.zeros((N, H + 6, W + 6, 4), dtype=torch.float32, device=dev))
            ops.conv_thin7(gz, self.mod.weight, View(gxp.buf, 0, self.mod.weight.shape[1]), pad=6, reflect=False, dgrad=True)      # gx_p[q] = sum_t W^T[t] gz[q - t]
            ops.unpad2d(gxp, gx4, 3, True)
            return
        g1 = View(new_act(N, H + 2, W + 2, gz.C, dev))
        ops.pad2d(gz, g1, 1, False)                                   # g1[u] = gz[u - 1]
        if want_w:
            for b in range(9):
                ops.wgrad_thin(g1, shifted[b], self.dws[b], self.db if b == 0 else None, flip=False, beta=0.0)
            self._gather_grad(True)
        if gx4 is None:
            return
        gxp = View(new_act(N, H + 6, W + 6, 4, dev))
        for b, (by, bx) in enumerate(_K7_BLOCKS):
            d = View(new_act(N, H + 2, W + 2, 4, dev))
            ops.conv_thin(g1, self.ws[b], d, dgrad=True)              # d[u] = sum_t W_b^T[t] gz[u - t]
            ops.window2d(d, gxp, -by, -bx, acc=b > 0)                 # gx_p[q] += d[q - b]
        ops.unpad2d(gxp, gx4, 3, True)

    # ---- features -> image (last layer): x [N,H,W,I] -> o4 [N,H,W,4] (no bias: it joins the tanh pass)
    def fwd_out(self, x, o4):
        N, H, W, dev = x.N, x.H, x.W, x.buf.device
        if self.fused:
            ops.conv_thin7(x, self.mod.weight, View(o4.buf, o4.coff, self.mod.weight.shape[0]), pad=3, reflect=True)     # (<= 3 image channels: the register-resident form)
            return x                                                  # (the un-padded input: the weight gradient reflects its reads)
        xp = View(new_act(N, H + 6, W + 6, x.C, dev))
        ops.pad2d(x, xp, 3, True)
        for b, (by, bx) in enumerate(_K7_BLOCKS):
            ob = View(new_act(N, H + 6, W + 6, 4, dev))
            ops.conv_thin(xp, self.ws[b], ob)                         # ob[u] = sum_t W_b[t] x_p[u + t - 1]
            ops.window2d(ob, o4, by + 1, bx + 1, acc=b > 0)           # y[p] += ob[p + b + 1]
        return xp

    def bwd_out(self, xp, go4, want_w, gx):
        """go4: [N,H,W,4] gradient of the image-side output; gx: [N,H,W,I] gradient of the layer's input."""
        N, H, W, dev = go4.N, go4.H, go4.W, go4.buf.device
        if self.fused:
            if want_w:
                ops.wgrad_thin7(xp, go4, self.mod.weight.grad, None, flip=True, rpad=3, off=-6)      # (xp: the layer's input as fwd_out returned it)
                if self.mod.bias is not None:
                    db4 = torch.zeros(4, dtype=torch.float32, device=dev)
                    ops.bias_grad(go4, db4, beta=0.0)
                    self.mod.bias.grad.add_(db4[:self.mod.bias.numel()])
            gc = View(new_act(N, H + 6, W + 6, 4, dev))
            ops.pad2d(go4, gc, 3, False)                              # gc[u] = g[u - 3]
            gxp = View(new_act(N, H + 6, W + 6, gx.C, dev))
            ops.conv(gc, self.packer.get(self.i_k7), gxp, mode=ops.CONV_7x7_C4)      # gx_p[q] = sum_t W^T[t] g[q - t]
            ops.unpad2d(gxp, gx, 3, True)
            return
        gxp = View(new_act(N, H + 6, W + 6, gx.C, dev))
        for b, (by, bx) in enumerate(_K7_BLOCKS):
            sm = View(new_act(N, H + 6, W + 6, 4, dev))
            ops.window2d(go4, sm, -(by + 1), -(bx + 1))               # sm[u] = g[u - b - 1]
            if want_w:
                ops.wgrad_thin(xp, sm, self.dws[b], self.db if b == 0 else None, flip=True, beta=0.0)
            epi = {} if b == 0 else dict(r1=gxp, beta1=1.0)
            ops.conv(sm, self.packer.get(self.i_mma[b]), gxp, mode=ops.CONV_3x3_C4, **epi)     # gx_p[u] += sum_t W_b^T[t] g[u - t - b]
        if want_w:
            self._gather_grad(True)
        ops.unpad2d(gxp, gx, 3, True)


class ResnetGenerator(HipNet):
    def __init__(self, input_nc, output_nc, ngf=64, norm_type="batch", use_dropout=False, n_blocks=6, padding_type="reflect",
                 upsample_mode="deconv"):
        super().__init__()
        if norm_type in ("BN", "batch"):
            norm = "batch"
        elif norm_type in ("IN", "instance"):
            norm = "instance"
        else:
            raise NameError("Unknown norm layer")
        if use_dropout or padding_type != "reflect" or upsample_mode != "deconv":
            raise NotImplementedError("HIP ResnetGenerator implements reflect padding, deconv up-sampling, no dropout")
        if input_nc > 4 or output_nc > 4 or ngf % 8:
            raise NotImplementedError("HIP ResnetGenerator needs <= 4 image channels and ngf %% 8 == 0")
        self.input_nc, self.output_nc, self.ngf, self.norm, self.n_blocks = input_nc, output_nc, ngf, norm, n_blocks
        ub = norm == "instance"                                   # use_bias (ResNet_arch.py:47-50)
        m = [B.Marker("reflectpad3"), B.Conv2dHIP(input_nc, ngf, 7, 1, bias=ub), _norm_module(norm, ngf), B.Marker("act:relu")]
        for i in range(2):
            mult = 2 ** i
            m += [B.Conv2dHIP(ngf * mult, ngf * mult * 2, 3, 2, bias=ub), _norm_module(norm, ngf * mult * 2), B.Marker("act:relu")]
        for _ in range(n_blocks):
            m.append(ResnetBlock(ngf * 4, norm, ub))
        for i in range(2):
            mult = 2 ** (2 - i)
            m += [ConvTranspose2dHIP(ngf * mult, ngf * mult // 2, 3, 2, bias=ub), _norm_module(norm, ngf * mult // 2), B.Marker("act:relu")]
        m += [B.Marker("reflectpad3"), B.Conv2dHIP(ngf, output_nc, 7, 1, bias=True), B.Marker("tanh")]
        self.model = nn.Sequential(*m)
        self._init_engine()

    # ------------------------------------------------------------------ executors
    def _build_ops(self, packer):
        m, nb = self.model, self.n_blocks
        self._c_in, self._c_out = m[1], m[17 + nb]
        # the thin / taps-in-K kernels take 16-, 32- or 64-channel wide sides; other widths keep the generic kernels
        # (the image-side kernels take <= 3 image channels and 16 / 32 / 64 features; anything else: the generic kernels of csrc/gconv.hip)
        self._k7 = (_K7Image(self._c_in, packer, "in"), _K7Image(self._c_out, packer, "out")) if (self.ngf in (16, 32, 64) and self.input_nc <= 3 and self.output_nc <= 3) else None
        self._downs = [_Padded4x4(m[4].weight, packer, False), _Padded4x4(m[7].weight, packer, False)]
        self._down_mods = [m[4], m[7]]
        self._ups = [_Padded4x4(m[10 + nb].weight, packer, True), _Padded4x4(m[13 + nb].weight, packer, True)]
        self._up_mods = [m[10 + nb], m[13 + nb]]
        self._norms = {"in": m[2], "d0": m[5], "d1": m[8], "u0": m[11 + nb], "u1": m[14 + nb]}
        self._blocks = []
        for i in range(nb):
            cb = m[10 + i].conv_block
            self._blocks.append(((ConvOp(cb[1], packer, need_dgrad=True), cb[2]), (ConvOp(cb[5], packer, need_dgrad=True), cb[6])))
        self._ops = True
        self._ones = {}

    def _refresh_derived(self):
        for p in self._downs + self._ups + list(self._k7 or ()):
            p.refresh()

    # normalisation: BatchNorm2d (train) over the batch, or InstanceNorm2d = the same kernels per image with unit affine
    def _affine(self, C, dev):
        k = (C, str(dev))
        if k not in self._ones:
            self._ones[k] = (torch.ones(C, dtype=torch.float32, device=dev), torch.zeros(C, dtype=torch.float32, device=dev))
        return self._ones[k]

    def _norm_fwd(self, mod, z, y, act):
        dev, C = z.buf.device, z.C
        if self.norm == "batch":
            mean, inv = torch.empty(C, device=dev), torch.empty(C, device=dev)
            ops.bn_train_fwd(z, y, mod.weight, mod.bias, mod.running_mean, mod.running_var, mod.num_batches_tracked, mean, inv,
                             momentum=mod.momentum, eps=mod.eps, act=act, slope=0.0)
            return [(mean, inv)]
        mean, inv = torch.empty(z.N * C, device=dev), torch.empty(z.N * C, device=dev)
        ops.instnorm_fwd(z, y, mean, inv, eps=1e-5, act=act, slope=0.0)      # one statistics group per image, one set of launches
        return [(mean, inv)]

    def _norm_bwd(self, mod, stats, gy, y, z, gz, relu, want_w):
        ms = 0.0 if relu else 1.0                                 # ReLU' gate taken from y (slope 0), or no gate at all
        if self.norm == "batch":
            mean, inv = stats[0]
            ops.bn_train_bwd(gy, y, z, gz, mod.weight, mean, inv, dgamma=mod.weight.grad if want_w else None,
                             dbeta=mod.bias.grad if want_w else None, mslope=ms)
            return
        mean, inv = stats[0]
        ops.instnorm_bwd(gy, y, z, gz, mean, inv, mslope=ms)

    # ------------------------------------------------------------------ forward
    def engine_forward(self, x, save):
        N, Cc, H, W = x.shape
        if H % 4 or W % 4:
            raise ValueError("ResnetGenerator needs input sizes divisible by 4")
        dev, ngf = x.device, self.ngf
        R, NONE = ops.ACT_RELU, ops.ACT_NONE
        x4 = View(new_act(N, H, W, 4, dev))
        ops.nchw_to_nhwc(x, x4, Cpad=4)
        tape = {}
        z = View(new_act(N, H, W, ngf, dev))
        if self._k7:
            x_in = self._k7[0].fwd_in(x4, z)              # (the nine shifted images: the weight gradient reads them again)
        else:
            ops.gconv_fwd(x4, self._c_in.weight, z, bias=self._c_in.bias, stride=1, pad=3, reflect=True)
            x_in = x4
        a = View(new_act(N, H, W, ngf, dev))
        tape["in"] = (x_in, z, a, self._norm_fwd(self._norms["in"], z, a, R))
        cur = a
        for i, (p4, mod) in enumerate(zip(self._downs, self._down_mods)):
            z = View(new_act(N, cur.H // 2, cur.W // 2, mod.out_channels, dev))
            p4.conv(cur, z, bias=mod.bias)
            a = View(new_act(N, z.H, z.W, z.C, dev))
            tape["d%d" % i] = (cur, z, a, self._norm_fwd(self._norms["d%d" % i], z, a, R))
            cur = a
        blocks = []
        for (c1, n1), (c2, n2) in self._blocks:
            h, w_, C = cur.H, cur.W, cur.C
            z1 = View(new_act(N, h, w_, C, dev))
            c1.fwd(cur, z1, reflect=True)                          # ReflectionPad2d(1) + conv3x3: the stager reflects the borders
            h1 = View(new_act(N, h, w_, C, dev))
            s1 = self._norm_fwd(n1, z1, h1, R)
            z2 = View(new_act(N, h, w_, C, dev))
            c2.fwd(h1, z2, reflect=True)
            y2 = View(new_act(N, h, w_, C, dev))
            s2 = self._norm_fwd(n2, z2, y2, NONE)
            out = View(new_act(N, h, w_, C, dev))
            ops.add2(out, cur, y2)
            blocks.append((cur, z1, h1, s1, z2, y2, s2))
            cur = out
        for i, (p4, mod) in enumerate(zip(self._ups, self._up_mods)):
            z = View(new_act(N, cur.H * 2, cur.W * 2, mod.out_channels, dev))
            p4.conv_t(cur, z, bias=mod.bias)
            a = View(new_act(N, z.H, z.W, z.C, dev))
            tape["u%d" % i] = (cur, z, a, self._norm_fwd(self._norms["u%d" % i], z, a, R))
            cur = a
        o4 = new_act(N, H, W, 4, dev)
        if self._k7:
            cur = self._k7[1].fwd_out(cur, View(o4, 0, 4))             # (keeps the reflection-padded input for the weight gradient)
        else:
            ops.gconv_fwd(cur, self._c_out.weight, View(o4, 0, 4), bias=None, stride=1, pad=3, reflect=True)     # bias added with the tanh pass
        pre = torch.empty((N, self.output_nc, H, W), dtype=torch.float32, device=dev)
        ops.nhwc_to_nchw(View(o4, 0, self.output_nc), pre)
        pre += self._c_out.bias.detach().view(1, -1, 1, 1)
        out = torch.empty_like(pre)
        ops.tanh_fwd(pre, out)
        saved = dict(tape=tape, blocks=blocks, last_in=cur, out=out) if save else None
        return out, saved

    # ------------------------------------------------------------------ backward
    def engine_backward(self, sv, gout, need_input_grad, need_param_grad):
        Wg = need_param_grad
        gout = gout.contiguous()
        dev = gout.device
        tape, blocks, last_in, out = sv["tape"], sv["blocks"], sv["last_in"], sv["out"]
        N, _, H, W = gout.shape
        gpre = torch.empty_like(gout)
        ops.tanh_bwd(gout, out, gpre)
        g4 = View(new_act(N, H, W, 4, dev))
        ops.nchw_to_nhwc(gpre, g4, Cpad=4)
        go = View(g4.buf, 0, 4)
        co = self._c_out
        g = View(new_act(N, H, W, self.ngf, dev))
        if self._k7:
            self._k7[1].bwd_out(last_in, go, Wg, g)
        else:
            if Wg:
                dw4 = torch.zeros((4, co.in_channels, 7, 7), dtype=torch.float32, device=dev)
                db4 = torch.zeros(4, dtype=torch.float32, device=dev)
                ops.gconv_wgrad(last_in, go, dw4, db4, stride=1, pad=3, reflect=True, beta=0.0)
                co.weight.grad.add_(dw4[:self.output_nc])
                co.bias.grad.add_(db4[:self.output_nc])
            w4 = torch.zeros((4, co.in_channels, 7, 7), dtype=torch.float32, device=dev)
            w4[:self.output_nc].copy_(co.weight.detach())
            ops.gconv_dgrad(go, w4, g, stride=1, pad=3, reflect=True)
        # up-sampling stages
        for i in (1, 0):
            xin, z, a, st = tape["u%d" % i]
            p4, mod = self._ups[i], self._up_mods[i]
            gz = View(new_act(N, z.H, z.W, z.C, dev))
            self._norm_bwd(self._norms["u%d" % i], st, g, a, z, gz, True, Wg)
            if Wg:
                p4.wgrad(gz, xin)                                  # roles swapped: the virtual convolution maps large -> small
                if mod.bias is not None:
                    ops.bias_grad(gz, mod.bias.grad)
            g = View(new_act(N, xin.H, xin.W, xin.C, dev))
            p4.conv(gz, g)
        # residual blocks
        for bi in range(len(blocks) - 1, -1, -1):
            xin, z1, h1, s1, z2, y2, s2 = blocks[bi]
            (c1, n1), (c2, n2) = self._blocks[bi]
            h, w_, C = z1.H, z1.W, z1.C
            gz2 = View(new_act(N, h, w_, C, dev))
            self._norm_bwd(n2, s2, g, y2, z2, gz2, False, Wg)
            if Wg:
                c2.wgrad(h1, gz2, reflect=True)                     # weight gradient over the reflected input, on the image grid
            # data-gradient with respect to the PADDED input (the zero-embedded gradient through the 3x3 data-gradient kernel
            # on the (h + 2) x (w + 2) grid), folded back by the adjoint of the reflection
            gzp = View(new_act(N, h + 2, w_ + 2, C, dev))
            ops.pad2d(gz2, gzp, 1, False)
            ghp = View(new_act(N, h + 2, w_ + 2, C, dev))
            c2.dgrad(gzp, ghp)
            gh1 = View(new_act(N, h, w_, C, dev))
            ops.unpad2d(ghp, gh1, 1, True)
            gz1 = View(new_act(N, h, w_, C, dev))
            self._norm_bwd(n1, s1, gh1, h1, z1, gz1, True, Wg)
            if Wg:
                c1.wgrad(xin, gz1, reflect=True)
            ops.pad2d(gz1, gzp, 1, False)
            c1.dgrad(gzp, ghp)
            gx = View(new_act(N, h, w_, C, dev))
            ops.unpad2d(ghp, gx, 1, True)
            gsum = View(new_act(N, h, w_, C, dev))
            ops.add2(gsum, gx, g)                                   # skip connection
            g = gsum
        # down-sampling stages
        for i in (1, 0):
            xin, z, a, st = tape["d%d" % i]
            p4, mod = self._downs[i], self._down_mods[i]
            gz = View(new_act(N, z.H, z.W, z.C, dev))
            self._norm_bwd(self._norms["d%d" % i], st, g, a, z, gz, True, Wg)
            if Wg:
                p4.wgrad(xin, gz)
                if mod.bias is not None:
                    ops.bias_grad(gz, mod.bias.grad)
            g = View(new_act(N, xin.H, xin.W, xin.C, dev))
            p4.conv_t(gz, g)
        x4, z, a, st = tape["in"]
        gz = View(new_act(N, z.H, z.W, z.C, dev))
        self._norm_bwd(self._norms["in"], st, g, a, z, gz, True, Wg)
        ci = self._c_in
        if self._k7:
            gx4 = View(new_act(N, H, W, 4, dev)) if need_input_grad else None
            self._k7[0].bwd_in(x4, gz, Wg, gx4)                      # (x4: the shifted images saved by fwd_in)
            if not need_input_grad:
                return None
        else:
            if Wg:
                dwi = torch.zeros((ci.out_channels, 4, 7, 7), dtype=torch.float32, device=dev)
                dbi = torch.zeros(ci.out_channels, dtype=torch.float32, device=dev)
                ops.gconv_wgrad(x4, gz, dwi, dbi, stride=1, pad=3, reflect=True, beta=0.0)
                ci.weight.grad.add_(dwi[:, :self.input_nc])
                if ci.bias is not None:
                    ci.bias.grad.add_(dbi)
            if not need_input_grad:
                return None
            wi4 = torch.zeros((ci.out_channels, 4, 7, 7), dtype=torch.float32, device=dev)
            wi4[:, :self.input_nc].copy_(ci.weight.detach())
            gx4 = View(new_act(N, H, W, 4, dev))
            ops.gconv_dgrad(gz, wi4, gx4, stride=1, pad=3, reflect=True)
        gin = torch.empty((N, self.input_nc, H, W), dtype=torch.float32, device=dev)
        ops.nhwc_to_nchw(View(gx4.buf, 0, self.input_nc), gin)
        return gin
