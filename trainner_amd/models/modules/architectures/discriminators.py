"""Discriminator_VGG (size-adaptive VGG-style discriminator) on the MI355X engine.

Constructor, state_dict keys and arithmetic follow codes/models/modules/architectures/
discriminators.py:16-51: conv3(in->nf)+LReLU; conv4s2+BN+LReLU; then [conv3(c->min(2c,512))+BN+LReLU;
conv4s2+BN+LReLU] until 4x4; flatten (NCHW order); Linear(.,100)+LReLU; Linear(100,1).
BatchNorm runs in training mode with per-replica batch statistics and in-place running-stat updates
on every call (4 per G+D step), exactly as nn.BatchNorm2d does under nn.DataParallel's replica 0.

The k4 s2 convolutions run as 2x2 convolutions over the space-to-depth view, their data-gradient as
four parity-decomposed 2x2 convolutions (conv_tile.hip); conv bias is kept although BN cancels it
(the reference has it: block.py:214-256 default bias=True).
"""
import torch
import torch.nn as nn

from .... import ops
from ....engine import ConvOp, HipNet
from ....ops import View, new_act
from . import block as B


class Discriminator_VGG(HipNet):
    def __init__(self, size, in_nc, base_nf, norm_type="batch", act_type="leakyrelu", mode="CNA", convtype="Conv2D",
                 arch="ESRGAN"):
        super().__init__()
        if norm_type != "batch" or mode != "CNA" or convtype != "Conv2D":
            raise NotImplementedError("Discriminator_VGG option outside the ESRGAN recipe is not implemented by the HIP engine")
        if in_nc > 4 or base_nf % 4:
            raise NotImplementedError("HIP Discriminator_VGG needs <= 4 image channels and base_nf %% 4 == 0")
        self.size, self.in_nc = size, in_nc
        self.act, self.slope = B.act_code(act_type)
        blocks = [B.conv_block(in_nc, base_nf, 3, 1, None, act_type), B.conv_block(base_nf, base_nf, 4, 2, "batch", act_type)]
        cur, nc = size // 2, base_nf
        while cur > 4:
            out = nc * 2 if nc < 512 else nc
            blocks.append(B.conv_block(nc, out, 3, 1, "batch", act_type))
            blocks.append(B.conv_block(out, out, 4, 2, "batch", act_type))
            nc, cur = out, cur // 2
        self.features = B.flat_sequential(*blocks)
        self.final_nc, self.final_size = nc, cur
        hidden = 128 if arch == "PPON" else 100
        self.classifier = nn.Sequential(B.LinearHIP(nc * cur * cur, hidden), B.Marker("act:leakyrelu"), B.LinearHIP(hidden, 1))
        self._init_engine()

    def _build_ops(self, packer):
        layers = []   # (ConvOp, bn module or None)
        mods = list(self.features)
        i = 0
        while i < len(mods):
            conv = mods[i]
            bn = mods[i + 1] if isinstance(mods[i + 1], B.BatchNorm2dHIP) else None
            layers.append((ConvOp(conv, packer, need_dgrad=True), bn))
            i += 3 if bn is not None else 2
        self._ops = layers

    def engine_forward(self, x, save):
        N, Cc, S, _ = x.shape
        if S != self.size or x.shape[3] != self.size:
            raise ValueError("Discriminator_VGG built for %dx%d inputs, got %s" % (self.size, self.size, tuple(x.shape)))
        dev, sl, act = x.device, self.slope, self.act
        x4 = View(new_act(N, S, S, 4, dev))
        ops.nchw_to_nhwc(x, x4, Cpad=4)
        cur, acts, bn_stats = x4, [], []
        for conv, bn in self._ops:
            m = conv.mod
            Ho = cur.H // m.stride
            if bn is None:
                y = View(new_act(N, Ho, Ho, m.out_channels, dev))
                conv.fwd(cur, y, act=act, slope=sl)
                acts.append((cur, None, y, None, None))
            else:
                z = View(new_act(N, Ho, Ho, m.out_channels, dev))
                conv.fwd(cur, z)
                y = View(new_act(N, Ho, Ho, m.out_channels, dev))
                mean = torch.empty(m.out_channels, dtype=torch.float32, device=dev)
                invstd = torch.empty(m.out_channels, dtype=torch.float32, device=dev)
                st = torch.empty(2 * m.out_channels, dtype=torch.float64, device=dev) if (save and self.memoize) else None
                ops.bn_train_fwd(z, y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                 mean, invstd, momentum=bn.momentum, eps=bn.eps, act=act, slope=sl, stat64=st)
                acts.append((cur, z, y, mean, invstd))
                bn_stats.append((bn, st))
            cur = y
        feat = torch.empty((N, self.final_nc, cur.H, cur.W), dtype=torch.float32, device=dev)   # NCHW flatten order
        ops.nhwc_to_nchw(cur, feat)
        l0, l1 = self.classifier[0], self.classifier[2]
        hid = torch.empty((N, l0.out_features), dtype=torch.float32, device=dev)
        ops.linear_fwd(feat.view(N, -1), l0.weight, l0.bias, hid, act=ops.ACT_LRELU, slope=0.2)
        out = torch.empty((N, 1), dtype=torch.float32, device=dev)
        ops.linear_fwd(hid, l1.weight, l1.bias, out)
        saved = dict(acts=acts, feat=feat, hid=hid, last=cur, bn_stats=bn_stats) if save else None
        return out, saved

    def replay_forward_side_effects(self, saved):
        """A repeated training-mode forward over the same batch moves every BatchNorm's running statistics once more (with the
        same batch statistics) and counts one more batch: exactly what tnr_bn_replay_running applies."""
        for bn, st in saved["bn_stats"]:
            if st is None:
                raise RuntimeError("memoized forward without recorded BatchNorm statistics")
            ops.bn_replay_running(bn.running_mean, bn.running_var, bn.num_batches_tracked, st, momentum=bn.momentum)

    def engine_backward(self, sv, gout, need_input_grad, need_param_grad):
        W, sl = need_param_grad, self.slope
        gout = gout.contiguous()
        dev = gout.device
        acts, feat, hid, last = sv["acts"], sv["feat"], sv["hid"], sv["last"]
        N = feat.shape[0]
        l0, l1 = self.classifier[0], self.classifier[2]
        ghid = torch.empty_like(hid)
        ops.linear_bwd(hid, l1.weight, gout, None, gx=ghid, dw=l1.weight.grad if W else None,
                       db=l1.bias.grad if W else None)
        gfeat = torch.empty_like(feat)
        ops.linear_bwd(feat.view(N, -1), l0.weight, ghid, hid, gx=gfeat.view(N, -1), dw=l0.weight.grad if W else None,
                       db=l0.bias.grad if W else None, mslope=0.2)
        gy = View(new_act(N, last.H, last.W, last.C, dev))
        ops.nchw_to_nhwc(gfeat, gy, Cpad=last.C)
        sched = getattr(self, "_bucket_schedule", None) if W else None      # data-parallel gradient buckets (dp.py)
        if sched is not None:
            sched.mark_done(l0.weight)
        for li in range(len(self._ops) - 1, -1, -1):
            conv, bn = self._ops[li]
            xin, z, y, mean, invstd = acts[li]
            if bn is not None:
                gz = View(new_act(N, y.H, y.W, y.C, dev))
                ops.bn_train_bwd(gy, y, z, gz, bn.weight, mean, invstd, dgamma=bn.weight.grad if W else None,
                                 dbeta=bn.bias.grad if W else None, mslope=sl, beta=bn.bias)
            else:
                gz = gy                                         # already masked by the consumer's dgrad epilogue
            if W:
                conv.wgrad(View(xin.buf, 0, conv.mod.in_channels) if li == 0 else xin, gz)
                if sched is not None:
                    sched.mark_done(conv.mod.weight)      # this layer's conv + BN and everything after it are final
            if li == 0:
                if not need_input_grad:
                    return None
                gx4 = View(new_act(N, xin.H, xin.W, 4, dev))
                conv.dgrad(gz, gx4)
                gx = torch.empty((N, self.in_nc, xin.H, xin.W), dtype=torch.float32, device=dev)
                ops.nhwc_to_nchw(View(gx4.buf, 0, self.in_nc), gx)
                return gx
            prev_bn = self._ops[li - 1][1]
            gprev = View(new_act(N, xin.H, xin.W, xin.C, dev))
            if prev_bn is None:
                conv.dgrad(gz, gprev, mask=xin, m_slope=sl)     # LeakyReLU' of the non-BN first layer
            else:
                conv.dgrad(gz, gprev)                           # bn_train_bwd applies LeakyReLU' itself
            gy = gprev
        return None


class UNetDiscriminator(HipNet):
    """U-Net discriminator with per-pixel logits (Real-ESRGAN) on the MI355X engine.

    Constructor, state_dict keys (conv0 .. conv9; only conv0 and conv9 carry a bias) and arithmetic follow
    codes/models/modules/architectures/discriminators.py:686-779: conv3(in->nf)+LReLU; three conv4s2 (no bias)+LReLU
    down to H/8; three times [bilinear x2 (align_corners=False) -> conv3 (no bias) + LReLU -> + skip]; two more
    conv3 + LReLU; conv3(nf->1) with bias.  `spectral_norm` cannot be enabled from the options
    (options/defaults.py:378-382 forwards only input_nc / nf / skip_connection) and is refused here.

    Kernels: the k3 / k4s2 implicit-GEMM tiles (conv_tile.hip) with the image-side layers on the vector ALUs;
    bilinear x2 and its adjoint as one HBM pass each (elementwise.hip).  The decoder activations are kept BEFORE the
    skip add as well (y4, y5, y6): LeakyReLU' is gated by their sign, which the sum x_k = y_k + skip no longer shows.
    """

    def __init__(self, input_nc, nf=64, skip_connection=True, spectral_norm=False):
        super().__init__()
        if spectral_norm:
            raise NotImplementedError("spectral norm in UNetDiscriminator is not implemented by the HIP engine")
        if input_nc > 4 or nf % 4:
            raise NotImplementedError("HIP UNetDiscriminator needs <= 4 image channels and nf %% 4 == 0")
        self.input_nc, self.nf, self.skip_connection = input_nc, nf, bool(skip_connection)
        self.slope = 0.2
        C = B.Conv2dHIP
        self.conv0 = C(input_nc, nf, 3, 1)
        self.conv1 = C(nf, nf * 2, 4, 2, bias=False)
        self.conv2 = C(nf * 2, nf * 4, 4, 2, bias=False)
        self.conv3 = C(nf * 4, nf * 8, 4, 2, bias=False)
        self.conv4 = C(nf * 8, nf * 4, 3, 1, bias=False)
        self.conv5 = C(nf * 4, nf * 2, 3, 1, bias=False)
        self.conv6 = C(nf * 2, nf, 3, 1, bias=False)
        self.conv7 = C(nf, nf, 3, 1, bias=False)
        self.conv8 = C(nf, nf, 3, 1, bias=False)
        self.conv9 = C(nf, 1, 3, 1)
        self._init_engine()

    def _build_ops(self, packer):
        self._ops = [ConvOp(getattr(self, "conv%d" % i), packer, need_dgrad=True) for i in range(10)]

    def engine_forward(self, x, save):
        N, Cc, H, W = x.shape
        if H % 8 or W % 8:
            raise ValueError("UNetDiscriminator needs input sizes divisible by 8, got %s" % (tuple(x.shape),))
        dev, sl, o, nf = x.device, self.slope, self._ops, self.nf
        act = dict(act=ops.ACT_LRELU, slope=sl)
        x4 = View(new_act(N, H, W, 4, dev))
        ops.nchw_to_nhwc(x, x4, Cpad=4)
        x0 = View(new_act(N, H, W, nf, dev))
        o[0].fwd(x4, x0, **act)
        x1 = View(new_act(N, H // 2, W // 2, nf * 2, dev))
        o[1].fwd(x0, x1, **act)
        x2 = View(new_act(N, H // 4, W // 4, nf * 4, dev))
        o[2].fwd(x1, x2, **act)
        x3 = View(new_act(N, H // 8, W // 8, nf * 8, dev))
        o[3].fwd(x2, x3, **act)
        # decoder: up(prev) -> conv + LReLU = y_k;  x_k = y_k + skip
        ups, ys, xs = [], [], []
        prev = x3
        for conv, skip in ((o[4], x2), (o[5], x1), (o[6], x0)):
            up = View(new_act(N, prev.H * 2, prev.W * 2, prev.C, dev))
            ops.bilinear2x_fwd(prev, up)
            y = View(new_act(N, up.H, up.W, conv.mod.out_channels, dev))
            conv.fwd(up, y, **act)
            if self.skip_connection:
                xk = View(new_act(N, up.H, up.W, y.C, dev))
                ops.add2(xk, y, skip)
            else:
                xk = y
            ups.append(up)
            ys.append(y)
            xs.append(xk)
            prev = xk
        o7 = View(new_act(N, H, W, nf, dev))
        o[7].fwd(prev, o7, **act)
        o8 = View(new_act(N, H, W, nf, dev))
        o[8].fwd(o7, o8, **act)
        l4 = new_act(N, H, W, 4, dev)
        o[9].fwd(o8, View(l4, 0, 1))
        out = torch.empty((N, 1, H, W), dtype=torch.float32, device=dev)
        ops.nhwc_to_nchw(View(l4, 0, 1), out)
        saved = dict(x4=x4, enc=(x0, x1, x2, x3), ups=ups, ys=ys, xs=xs, o7=o7, o8=o8) if save else None
        return out, saved

    def engine_backward(self, sv, gout, need_input_grad, need_param_grad):
        Wg, sl, o, nf = need_param_grad, self.slope, self._ops, self.nf
        gout = gout.contiguous()
        dev = gout.device
        x4 = sv["x4"]
        x0, x1, x2, x3 = sv["enc"]
        ups, ys, xs, o7, o8 = sv["ups"], sv["ys"], sv["xs"], sv["o7"], sv["o8"]
        N, H, W = x4.N, x4.H, x4.W
        sched = getattr(self, "_bucket_schedule", None) if Wg else None      # data-parallel gradient buckets (dp.py)

        def done(i):
            if sched is not None:
                sched.mark_done(o[i].mod.weight)

        g4 = new_act(N, H, W, 4, dev)
        ops.nchw_to_nhwc(gout, View(g4), Cpad=4)
        g9 = View(g4, 0, 1)
        if Wg:
            o[9].wgrad(o8, g9)
            done(9)
        g8 = View(new_act(N, H, W, nf, dev))
        o[9].dgrad(g9, g8, mask=o8, m_slope=sl)
        if Wg:
            o[8].wgrad(o7, g8)
            done(8)
        g7 = View(new_act(N, H, W, nf, dev))
        o[8].dgrad(g8, g7, mask=o7, m_slope=sl)
        if Wg:
            o[7].wgrad(xs[2], g7)
            done(7)
        gx = View(new_act(N, H, W, nf, dev))                 # gradient w.r.t. x6 = y6 + x0: feeds y6 (gated) and the skip
        o[7].dgrad(g7, gx)
        skips = []                                           # skip gradients for x0, x1, x2 (in that order)
        enc_in = (x2, x1, x0)
        for lvl in (2, 1, 0):                                # conv6, conv5, conv4
            conv, up, y = o[4 + lvl], ups[lvl], ys[lvl]
            if lvl == 2:
                gz = View(new_act(N, y.H, y.W, y.C, dev))
                ops.mask_copy(gz, gx, y, sl)                 # gz = gx * LeakyReLU'(y6)
            else:
                gz = gnext_z                                 # produced by the bilinear adjoint below
            skips.append(gx if self.skip_connection else None)
            if Wg:
                conv.wgrad(up, gz)
                done(4 + lvl)
            gup = View(new_act(N, up.H, up.W, up.C, dev))
            conv.dgrad(gz, gup)
            # adjoint of the up-sampling: plain gradient of the tensor that was up-sampled (x5 / x4: a skip sum; x3: an
            # encoder activation) and its LeakyReLU'-gated copy for the producing convolution
            src_y = ys[lvl - 1] if lvl > 0 else x3
            gnext_z = View(new_act(N, up.H // 2, up.W // 2, up.C, dev))
            if lvl > 0 and self.skip_connection:
                gx = View(new_act(N, up.H // 2, up.W // 2, up.C, dev))
                ops.bilinear2x_bwd(gup, gx=gx, gz=gnext_z, mask=src_y, mslope=sl)
            else:
                gx = None
                ops.bilinear2x_bwd(gup, gx=None, gz=gnext_z, mask=src_y, mslope=sl)
        # encoder: conv3 <- gz3; each data-gradient adds the skip gradient of its input and gates by its LeakyReLU'
        gz = gnext_z                                         # gradient w.r.t. conv3's pre-activation
        s_x0, s_x1, s_x2 = skips[0], skips[1], skips[2]
        for i, xin, skip in ((3, x2, s_x2), (2, x1, s_x1), (1, x0, s_x0)):
            if Wg:
                o[i].wgrad(xin, gz)
                done(i)
            gprev = View(new_act(N, xin.H, xin.W, xin.C, dev))
            kw = dict(r1=skip, beta1=1.0) if skip is not None else {}
            o[i].dgrad(gz, gprev, mask=xin, m_slope=sl, **kw)
            gz = gprev
        if Wg:
            o[0].wgrad(View(x4.buf, 0, self.input_nc), gz)
        if not need_input_grad:
            return None
        gx4 = View(new_act(N, H, W, 4, dev))
        o[0].dgrad(gz, gx4)
        gin = torch.empty((N, self.input_nc, H, W), dtype=torch.float32, device=dev)
        ops.nhwc_to_nchw(View(gx4.buf, 0, self.input_nc), gin)
        return gin


class _PadCinOp:
    """A convolution whose input-channel count is not a multiple of 4 (the conditional PatchGAN's first layer: 6 = A + B
    channels) on the matrix-core kernels: the NHWC input buffer already carries the channels zero-extended to a multiple of 4,
    the weight gets the same zero extension (a derived tensor, re-filled whenever the parameters change), and the weight
    gradient comes back through it."""

    def __init__(self, conv, packer, cin_buf):
        self.conv = conv
        shadow = nn.Module()
        shadow.weight = nn.Parameter(torch.zeros((conv.out_channels, cin_buf, conv.kernel_size, conv.kernel_size),
                                                 dtype=torch.float32, device=conv.weight.device), requires_grad=False)
        shadow.weight.grad = torch.zeros_like(shadow.weight)
        shadow.bias = conv.bias
        shadow.kernel_size, shadow.stride = conv.kernel_size, conv.stride
        shadow.in_channels, shadow.out_channels = cin_buf, conv.out_channels
        self.shadow = shadow
        self.op = ConvOp(shadow, packer, need_dgrad=True)

    def refresh(self):
        self.shadow.weight.data[:, :self.conv.in_channels].copy_(self.conv.weight.detach())

    def fwd(self, x, y, **epi):
        self.op.fwd(x, y, **epi)

    def dgrad(self, g, gx):
        self.op.dgrad(g, gx)

    def wgrad(self, x, g):
        ops.fill(self.shadow.weight.grad, 0.0)
        self.op.wgrad(x, g)
        self.conv.weight.grad.add_(self.shadow.weight.grad[:, :self.conv.in_channels])


class _K4S1:
    """A 4x4 stride-1 pad-1 convolution (the PatchGAN's last two layers) on the matrix cores.
    Forward and data-gradient go through the patch matrix (tnr_im2col + the 1x1 implicit-GEMM kernel: ops.conv_col; the
    data-gradient of a stride-1 layer is the same convolution with pad 2 and flipped / transposed taps).  The weight
    gradient is taken as four shifted 3x3 windows of the MFMA weight-gradient kernel: with G_s = the output gradient
    zero-embedded at offset s in {0,1}^2 of the input grid, wgrad3x3(x, G_s)[ty][tx] = dW[ty + sy][tx + sx], and the four
    3x3 blocks cover the 4x4 taps (one grouped launch).  The output-channel dimension is zero-extended to the buffer's
    multiple of 4 (the 512 -> 1 logit layer)."""

    def __init__(self, conv, packer, cin_buf):
        self.mod, self.packer = conv, packer
        w = conv.weight
        self.co = (w.shape[0] + 3) // 4 * 4
        self.alias = self.co == w.shape[0] and cin_buf == w.shape[1]
        self.wp4 = w if self.alias else torch.zeros((self.co, cin_buf, 4, 4), dtype=torch.float32, device=w.device)
        self.i_f = packer.add(self.wp4, ops.PACK_COL_FWD)
        self.i_d = packer.add(self.wp4, ops.PACK_COL_DGRAD3)
        self.bias = None
        if conv.bias is not None:
            self.bias = conv.bias if self.co == conv.out_channels else torch.zeros(self.co, dtype=torch.float32, device=w.device)
        self.dw3 = torch.zeros((4, self.co, cin_buf, 3, 3), dtype=torch.float32, device=w.device)
        self.db = torch.zeros(self.co, dtype=torch.float32, device=w.device)

    def refresh(self):
        w = self.mod.weight.detach()
        if not self.alias:
            self.wp4[:w.shape[0], :w.shape[1]].copy_(w)
        if self.bias is not None and self.bias is not self.mod.bias:
            self.bias[:self.mod.out_channels].copy_(self.mod.bias.detach())

    def fwd(self, x, y, **epi):
        ops.conv_col(x, self.packer.get(self.i_f), y, 4, 1, 1, bias=self.bias, **epi)

    def dgrad(self, g, gx):
        ops.conv_col(g, self.packer.get(self.i_d), gx, 4, 1, 2)

    def wgrad(self, x, g):
        items = []
        for s in range(4):
            gs = View(new_act(x.N, x.H, x.W, g.C, x.buf.device))
            ops.window2d(g, gs, -(s >> 1), -(s & 1))
            items.append(dict(x=x, g=gs, dw=self.dw3[s], db=self.db if s == 0 else None, beta=0.0))
        ops.wgrad_group(items, mode=ops.CONV_3x3)
        O, I = self.mod.out_channels, self.mod.in_channels
        gw, d = self.mod.weight.grad, self.dw3
        gw[:, :, :3, :3].add_(d[0, :O, :I])                       # s = (0, 0): taps [0, 3)^2
        gw[:, :, 3, 1:].add_(d[3, :O, :I, 2, :])                  # s = (1, 1): ky = 3, kx = 1..3
        gw[:, :, 1:3, 3].add_(d[3, :O, :I, :2, 2])                #             ky = 1..2, kx = 3
        gw[:, :, 3, 0].add_(d[2, :O, :I, 2, 0])                   # s = (1, 0): tap (3, 0)
        gw[:, :, 0, 3].add_(d[1, :O, :I, 0, 2])                   # s = (0, 1): tap (0, 3)
        if self.mod.bias is not None:
            self.mod.bias.grad.add_(self.db[:O])


class NLayerDiscriminator(HipNet):
    """PatchGAN discriminator (Pix2Pix / CycleGAN) on the MI355X engine.

    Constructor, `state_dict` keys (`model.<i>.*`) and arithmetic follow codes/models/modules/architectures/
    discriminators.py:472-579 for the default configuration get_network builds (BatchNorm2d, patch output, no spectral norm,
    no intermediate feature maps): conv4 s2 (in -> ndf, bias) + LReLU; n_layers-1 x [conv4 s2 (no bias) + BN + LReLU];
    conv4 s1 (no bias) + BN + LReLU; conv4 s1 (-> 1, bias).  The stride-2 layers run on the space-to-depth MFMA kernels
    (conv_tile.hip), the two stride-1 4x4 layers through the patch matrix + 1x1 GEMM / shifted 3x3 weight-gradient windows
    (_K4S1); only a first layer with a channel count that is not a multiple of 4 would use the generic kernel (csrc/gconv.hip)."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=None, use_sigmoid=False, get_feats=False, patch=True,
                 use_spectral_norm=False):
        super().__init__()
        if use_sigmoid or get_feats or not patch or use_spectral_norm or norm_layer is not None:
            raise NotImplementedError("HIP NLayerDiscriminator implements the default PatchGAN (BatchNorm, patch output)")
        if input_nc > 8 or ndf % 8:
            raise NotImplementedError("HIP NLayerDiscriminator needs <= 8 input channels and ndf %% 8 == 0")
        self.input_nc, self.n_layers, self.slope = input_nc, n_layers, 0.2
        seq = [B.Conv2dHIP(input_nc, ndf, 4, 2), B.Marker("act:leakyrelu")]
        nf_mult = 1
        for n in range(1, n_layers):
            prev, nf_mult = nf_mult, min(2 ** n, 8)
            seq += [B.Conv2dHIP(ndf * prev, ndf * nf_mult, 4, 2, bias=False), B.BatchNorm2dHIP(ndf * nf_mult), B.Marker("act:leakyrelu")]
        prev, nf_mult = nf_mult, min(2 ** n_layers, 8)
        seq += [B.Conv2dHIP(ndf * prev, ndf * nf_mult, 4, 1, bias=False), B.BatchNorm2dHIP(ndf * nf_mult), B.Marker("act:leakyrelu")]
        seq += [B.Conv2dHIP(ndf * nf_mult, 1, 4, 1)]
        self.model = nn.Sequential(*seq)
        self._init_engine()

    def _build_ops(self, packer):
        mods = list(self.model)
        self._layers = []          # (conv module, bn or None, has activation, ConvOp for the stride-2 layers)
        i = 0
        while i < len(mods):
            conv = mods[i]
            bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], B.BatchNorm2dHIP) else None
            j = i + (2 if bn is not None else 1)
            act = j < len(mods) and isinstance(mods[j], B.Marker)
            if conv.in_channels % 4 == 0:
                op = ConvOp(conv, packer, need_dgrad=True) if conv.stride == 2 else _K4S1(conv, packer, conv.in_channels)
            else:
                op = _PadCinOp(conv, packer, (conv.in_channels + 3) // 4 * 4) if conv.stride == 2 else None
            self._layers.append((conv, bn, act, op))
            i = j + (1 if act else 0)
        self._ops = True

    def _refresh_derived(self):
        for _, _, _, op in self._layers:
            if isinstance(op, (_K4S1, _PadCinOp)):
                op.refresh()

    def engine_forward(self, x, save):
        N, Cc, H, W = x.shape
        dev, sl = x.device, self.slope
        cpad = (Cc + 3) // 4 * 4
        xin = View(new_act(N, H, W, cpad, dev))
        ops.nchw_to_nhwc(x, xin, Cpad=cpad)
        cur, tape = xin, []
        for li, (conv, bn, act, op) in enumerate(self._layers):
            k, s = conv.kernel_size, conv.stride
            Ho, Wo = (cur.H + 2 - k) // s + 1, (cur.W + 2 - k) // s + 1
            co = (conv.out_channels + 3) // 4 * 4
            z = View(new_act(N, Ho, Wo, co, dev))
            fused_act = act and bn is None
            if op is not None:
                op.fwd(cur, z, **(dict(act=ops.ACT_LRELU, slope=sl) if fused_act else {}))
            else:
                ops.gconv_fwd(cur, self._wpad(conv, cur.C), z, bias=self._bpad(conv, co),
                              stride=s, pad=1, reflect=False, act=ops.ACT_LRELU if fused_act else ops.ACT_NONE, slope=sl)
            y, stats = z, None
            if bn is not None:
                y = View(new_act(N, Ho, Wo, co, dev))
                mean, inv = torch.empty(co, device=dev), torch.empty(co, device=dev)
                ops.bn_train_fwd(z, y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, mean, inv,
                                 momentum=bn.momentum, eps=bn.eps, act=ops.ACT_LRELU if act else ops.ACT_NONE, slope=sl)
                stats = (mean, inv)
            tape.append((cur, z, y, stats))
            cur = y
        out = torch.empty((N, 1, cur.H, cur.W), dtype=torch.float32, device=dev)
        ops.nhwc_to_nchw(View(cur.buf, 0, 1), out)
        return out, (dict(tape=tape) if save else None)

    def _wpad(self, conv, cin_buf):
        """OIHW weight with the input-channel dim zero-extended to the buffer's channel count and the output dim to a multiple of 4
        (generic kernel reads `Cin` = buffer channels)."""
        w = conv.weight.detach()
        co = (w.shape[0] + 3) // 4 * 4
        if w.shape[1] == cin_buf and co == w.shape[0]:
            return w
        wp = torch.zeros((co, cin_buf, w.shape[2], w.shape[3]), dtype=torch.float32, device=w.device)
        wp[:w.shape[0], :w.shape[1]].copy_(w)
        return wp

    def _bpad(self, conv, co):
        if conv.bias is None:
            return None
        if co == conv.out_channels:
            return conv.bias.detach()
        b = torch.zeros(co, dtype=torch.float32, device=conv.bias.device)
        b[:conv.out_channels].copy_(conv.bias.detach())
        return b

    def engine_backward(self, sv, gout, need_input_grad, need_param_grad):
        Wg, sl = need_param_grad, self.slope
        gout = gout.contiguous()
        dev = gout.device
        tape = sv["tape"]
        N = gout.shape[0]
        last = tape[-1][2]
        gy = View(new_act(N, last.H, last.W, last.C, dev))
        ops.nchw_to_nhwc(gout, gy, Cpad=last.C)
        sched = getattr(self, "_bucket_schedule", None) if Wg else None
        for li in range(len(self._layers) - 1, -1, -1):
            conv, bn, act, op = self._layers[li]
            xin, z, y, stats = tape[li]
            if bn is not None:
                gz = View(new_act(N, z.H, z.W, z.C, dev))
                ops.bn_train_bwd(gy, y, z, gz, bn.weight, stats[0], stats[1], dgamma=bn.weight.grad if Wg else None,
                                 dbeta=bn.bias.grad if Wg else None, mslope=sl if act else 1.0)
            elif act:
                gz = gy
                ops.mask_mul(gz, y, sl)                              # LeakyReLU' of a layer without BatchNorm
            else:
                gz = gy
            k, s = conv.kernel_size, conv.stride
            if Wg:
                if op is not None:
                    op.wgrad(xin, gz)
                else:
                    co = gz.C
                    dwp = torch.zeros((co, xin.C, k, k), dtype=torch.float32, device=dev)
                    dbp = torch.zeros(co, dtype=torch.float32, device=dev)
                    ops.gconv_wgrad(xin, gz, dwp, dbp, stride=s, pad=1, reflect=False, beta=0.0)
                    conv.weight.grad.add_(dwp[:conv.out_channels, :conv.in_channels])
                    if conv.bias is not None:
                        conv.bias.grad.add_(dbp[:conv.out_channels])
                if sched is not None:
                    sched.mark_done(conv.weight)
            if li == 0 and not need_input_grad:
                return None
            gx = View(new_act(N, xin.H, xin.W, xin.C, dev))
            if op is not None:
                op.dgrad(gz, gx)
            else:
                ops.gconv_dgrad(gz, self._wpad(conv, xin.C), gx, stride=s, pad=1, reflect=False)
            gy = gx
        gin = torch.empty((N, self.input_nc, gy.H, gy.W), dtype=torch.float32, device=dev)
        ops.nhwc_to_nchw(View(gy.buf, 0, self.input_nc), gin)
        return gin
