"""UnetGenerator (the Pix2Pix generator of the shipped recipe, options/i2i/train_pix2pix.yml:65) on the MI355X engine.

Constructor, `state_dict` keys and arithmetic follow codes/models/modules/architectures/UNet_arch.py:11-162: `num_downs` nested
UnetSkipConnectionBlocks, built from the innermost outwards.  A block maps x to cat[x, up(sub(down(x)))] with
    down = LeakyReLU(0.2) -> Conv2d(k4, s2, p1) -> norm          (outermost: no activation, no norm; innermost: no norm)
    up   = ReLU -> ConvTranspose2d(k4, s2, p1) -> norm           (outermost: bias, Tanh instead of norm)
norm = BatchNorm2d (train mode; convolutions then carry no bias) or InstanceNorm2d (no affine; convolutions carry a bias).  Both
activations are IN PLACE in the reference (:106,108), so the tensor a block concatenates is the ACTIVATED x: with e_k the k-th
encoder output, every consumer sees a_k = LeakyReLU(e_k) -- the next encoder convolution directly, the decoder through the parent's ReLU
as ReLU(a_k) = ReLU(e_k).  The engine therefore never stores e_k: the normalisation kernels write a_k (LeakyReLU fused), one pass
writes ReLU(a_k) into the first half of the decoder's concatenation buffer and the decoder's normalisation writes ReLU(d_{k+1}) into
the second half -- no torch.cat.

Kernel mapping (all fp32 NHWC): every convolution is the 4x4 stride-2 space-to-depth MFMA kernel; a ConvTranspose2d(k4, s2, p1) IS the
data-gradient of that geometry (TNR_DGRAD_4x4_S2) with its [in, out, 4, 4] weight read as a convolution weight [O = in, I = out], its
input gradient is the forward kernel and its weight gradient the weight-gradient kernel with the roles of the two tensors swapped
(engine.ConvOp on a shim module).  The 3-channel image sides are zero-extended to 4 channels (weights and buffers).  Layers of <= 4096
output pixels go through the im2col + split-K GEMM (ops.small_gemm_ok), as in the discriminator's tail.
"""
import torch
import torch.nn as nn

from .... import ops
from ....engine import ConvOp, HipNet
from ....ops import View, new_act
from . import block as B
from .ResNet_arch import ConvTranspose2dHIP


class UnetSkipConnectionBlock(nn.Module):
    """Parameter holder with the reference's child indices (UNet_arch.py:72-162); executed by UnetGenerator's engine."""

    def __init__(self, outer_nc, inner_nc, input_nc=None, submodule=None, outermost=False, innermost=False, norm="batch",
                 use_dropout=False, upsample_mode="deconv"):
        super().__init__()
        if upsample_mode != "deconv":
            raise NotImplementedError("HIP UnetGenerator implements the original deconv up-sampling")
        self.outermost, self.innermost = outermost, innermost
        use_bias = norm == "instance"
        input_nc = outer_nc if input_nc is None else input_nc
        downconv = B.Conv2dHIP(input_nc, inner_nc, 4, 2, bias=use_bias)

        def norm_mod(nc):
            return B.BatchNorm2dHIP(nc, affine=True) if norm == "batch" else B.Marker("instancenorm")

        if outermost:
            model = [downconv, submodule, B.Marker("act:relu"), ConvTranspose2dHIP(inner_nc * 2, outer_nc, 4, 2, bias=True), B.Marker("tanh")]
        elif innermost:
            model = [B.Marker("act:leakyrelu"), downconv, B.Marker("act:relu"), ConvTranspose2dHIP(inner_nc, outer_nc, 4, 2, bias=use_bias),
                     norm_mod(outer_nc)]
        else:
            model = [B.Marker("act:leakyrelu"), downconv, norm_mod(inner_nc), submodule, B.Marker("act:relu"),
                     ConvTranspose2dHIP(inner_nc * 2, outer_nc, 4, 2, bias=use_bias), norm_mod(outer_nc)]
            if use_dropout:
                model.append(B.Marker("dropout0.5"))
        self.model = nn.Sequential(*model)


class _TConv:
    """ConvTranspose2d(k4, s2, p1) of the decoder on the 4x4 stride-2 kernels: its weight [in, out, 4, 4] read as the convolution
    weight [O = in, I = out] whose data-gradient it is; output channel counts that are not a multiple of 4 (the image) are
    zero-extended through a shadow copy (re-filled whenever the parameters change; the gradient comes back through it)."""

    def __init__(self, mod, packer):
        self.mod = mod
        O, I = mod.weight.shape[0], mod.weight.shape[1]
        self.ipad = (I + 3) // 4 * 4
        shim = nn.Module()
        if self.ipad == I:
            shim.weight = mod.weight                      # same tensor, same gradient view
        else:
            shim.weight = nn.Parameter(torch.zeros((O, self.ipad, 4, 4), dtype=torch.float32, device=mod.weight.device), requires_grad=False)
            shim.weight.grad = torch.zeros_like(shim.weight)
        shim.bias, shim.kernel_size, shim.stride, shim.in_channels, shim.out_channels = None, 4, 2, self.ipad, O
        self.shim = shim
        self.op = ConvOp(shim, packer, need_dgrad=True)
        self.bias_pad = None

    def refresh(self):
        I = self.mod.weight.shape[1]
        if self.ipad != I:
            self.shim.weight.data[:, :I].copy_(self.mod.weight.detach())
            if self.mod.bias is not None:
                if self.bias_pad is None:
                    self.bias_pad = torch.zeros(self.ipad, dtype=torch.float32, device=self.mod.weight.device)
                self.bias_pad[:I].copy_(self.mod.bias.detach())

    def bias(self):
        return self.bias_pad if self.ipad != self.mod.weight.shape[1] else self.mod.bias

    def fwd(self, x_small, y_large, **epi):               # the transposed convolution = the data-gradient kernel
        self.op.dgrad(x_small, y_large, bias=self.bias(), **epi)

    def bwd_data(self, g_large, g_small, **epi):          # its input gradient = the forward kernel
        self.op.fwd(g_large, g_small, **epi)

    def wgrad(self, g_large, x_small):
        I = self.mod.weight.shape[1]
        if self.ipad != I:
            ops.fill(self.shim.weight.grad, 0.0)
        self.op.wgrad(g_large, x_small, with_bias=False)
        if self.ipad != I:
            self.mod.weight.grad.add_(self.shim.weight.grad[:, :I])
        if self.mod.bias is not None:
            if self.ipad != I:
                db = torch.zeros(self.ipad, dtype=torch.float32, device=g_large.buf.device)
                ops.bias_grad(g_large, db, beta=0.0)
                self.mod.bias.grad.add_(db[:I])
            else:
                ops.bias_grad(g_large, self.mod.bias.grad)


class _DConv:
    """Conv2d(k4, s2, p1) of the encoder; an input channel count that is not a multiple of 4 (the image) is zero-extended."""

    def __init__(self, mod, packer):
        self.mod = mod
        O, I = mod.weight.shape[0], mod.weight.shape[1]
        self.ipad = (I + 3) // 4 * 4
        if self.ipad == I:
            self.shim = mod
        else:
            shim = nn.Module()
            shim.weight = nn.Parameter(torch.zeros((O, self.ipad, 4, 4), dtype=torch.float32, device=mod.weight.device), requires_grad=False)
            shim.weight.grad = torch.zeros_like(shim.weight)
            shim.bias, shim.kernel_size, shim.stride, shim.in_channels, shim.out_channels = mod.bias, 4, 2, self.ipad, O
            self.shim = shim
        self.op = ConvOp(self.shim, packer, need_dgrad=True)

    def refresh(self):
        if self.shim is not self.mod:
            self.shim.weight.data[:, :self.mod.weight.shape[1]].copy_(self.mod.weight.detach())

    def fwd(self, x, y, **epi):
        self.op.fwd(x, y, **epi)

    def dgrad(self, g, gx, **epi):
        self.op.dgrad(g, gx, **epi)

    def wgrad(self, x, g):
        if self.shim is not self.mod:
            ops.fill(self.shim.weight.grad, 0.0)
        self.op.wgrad(x, g)                                # (+ bias gradient: the shim shares the bias parameter)
        if self.shim is not self.mod:
            self.mod.weight.grad.add_(self.shim.weight.grad[:, :self.mod.weight.shape[1]])


class UnetGenerator(HipNet):
    def __init__(self, input_nc, output_nc, num_downs, ngf=64, norm_type="batch", use_dropout=False, upsample_mode="deconv"):
        super().__init__()
        if norm_type in ("BN", "batch"):
            norm = "batch"
        elif norm_type in ("IN", "instance"):
            norm = "instance"
        else:
            raise NameError("Unknown norm layer")
        if use_dropout:
            raise NotImplementedError("HIP UnetGenerator: dropout (stochastic, off in the reference's defaults) is not implemented")
        if num_downs < 5 or input_nc > 4 or output_nc > 4 or ngf % 8:
            raise NotImplementedError("HIP UnetGenerator needs num_downs >= 5, <= 4 image channels and ngf %% 8 == 0")
        self.input_nc, self.output_nc, self.num_downs, self.ngf, self.norm = input_nc, output_nc, num_downs, ngf, norm
        blk = UnetSkipConnectionBlock(ngf * 8, ngf * 8, innermost=True, norm=norm)
        for _ in range(num_downs - 5):
            blk = UnetSkipConnectionBlock(ngf * 8, ngf * 8, submodule=blk, norm=norm, use_dropout=use_dropout)
        blk = UnetSkipConnectionBlock(ngf * 4, ngf * 8, submodule=blk, norm=norm)
        blk = UnetSkipConnectionBlock(ngf * 2, ngf * 4, submodule=blk, norm=norm)
        blk = UnetSkipConnectionBlock(ngf, ngf * 2, submodule=blk, norm=norm)
        self.model = UnetSkipConnectionBlock(output_nc, ngf, input_nc=input_nc, submodule=blk, outermost=True, norm=norm)
        self._init_engine()

    # ------------------------------------------------------------------ executors
    def _levels(self):
        """[(downconv, downnorm or None, upconv, upnorm or None)] from the outermost block inwards."""
        out, blk = [], self.model
        while blk is not None:
            m = list(blk.model)
            if blk.outermost:
                out.append((m[0], None, m[3], None))
                blk = m[1]
            elif blk.innermost:
                out.append((m[1], None, m[3], m[4]))
                blk = None
            else:
                out.append((m[1], m[2], m[5], m[6]))
                blk = m[3]
        return out

    def _build_ops(self, packer):
        self._lv = [(_DConv(dc, packer), dn, _TConv(uc, packer), un) for dc, dn, uc, un in self._levels()]
        self._ops = True

    def _refresh_derived(self):
        for d, _, u, _ in self._lv:
            d.refresh()
            u.refresh()

    def _norm_fwd(self, mod, z, y, act, slope):
        dev, C = z.buf.device, z.C
        if self.norm == "batch":
            mean, inv = torch.empty(C, device=dev), torch.empty(C, device=dev)
            ops.bn_train_fwd(z, y, mod.weight, mod.bias, mod.running_mean, mod.running_var, mod.num_batches_tracked, mean, inv,
                             momentum=mod.momentum, eps=mod.eps, act=act, slope=slope)
        else:
            mean, inv = torch.empty(z.N * C, device=dev), torch.empty(z.N * C, device=dev)
            ops.instnorm_fwd(z, y, mean, inv, eps=1e-5, act=act, slope=slope)
        return mean, inv

    def _norm_bwd(self, mod, stats, gy, y, z, gz, mslope, want_w):
        if self.norm == "batch":
            ops.bn_train_bwd(gy, y, z, gz, mod.weight, stats[0], stats[1], dgamma=mod.weight.grad if want_w else None,
                             dbeta=mod.bias.grad if want_w else None, mslope=mslope)
        else:
            ops.instnorm_bwd(gy, y, z, gz, stats[0], stats[1], mslope=mslope)

    # ------------------------------------------------------------------ forward
    def engine_forward(self, x, save):
        N, Cc, H, W = x.shape
        L = self.num_downs
        if H % (1 << L) or W % (1 << L):
            raise ValueError("UnetGenerator with %d down-samplings needs input sizes divisible by %d" % (L, 1 << L))
        dev = x.device
        LR, RE = ops.ACT_LRELU, ops.ACT_RELU
        cpad = (Cc + 3) // 4 * 4
        xin = View(new_act(N, H, W, cpad, dev))
        ops.nchw_to_nhwc(x, xin, Cpad=cpad)
        lv = self._lv
        # ---- encoder: a[i] = LeakyReLU(e_i) (innermost: ReLU(e) -- its only consumer is the decoder's ReLU)
        a, zs, dstats = [], [], []
        cur = xin
        for i, (dc, dn, _uc, _un) in enumerate(lv):
            co = dc.mod.out_channels
            if i == 0 or i == L - 1:
                y = View(new_act(N, cur.H // 2, cur.W // 2, co, dev))
                dc.fwd(cur, y, act=LR if i == 0 else RE, slope=0.2 if i == 0 else 0.0)
                zs.append(None)
                dstats.append(None)
            else:
                z = View(new_act(N, cur.H // 2, cur.W // 2, co, dev))
                dc.fwd(cur, z)
                y = View(new_act(N, z.H, z.W, co, dev))
                dstats.append(self._norm_fwd(dn, z, y, LR, 0.2))
                zs.append(z)
            a.append(y)
            cur = y
        # ---- decoder: U[i] = [ReLU(a_i) | ReLU(norm(convT_{i+1}(.)))] is the input of level i's transposed convolution
        U, ts, ustats = [None] * L, [None] * L, [None] * L
        src = a[L - 1]                                   # ReLU(e_innermost)
        for i in range(L - 1, 0, -1):
            uc, un = lv[i][2], lv[i][3]
            co = uc.mod.out_channels                      # = channels of a[i - 1]
            t = View(new_act(N, src.H * 2, src.W * 2, co, dev))
            uc.fwd(src, t)
            buf = new_act(N, t.H, t.W, 2 * co, dev)
            ops.mask_copy(View(buf, 0, co), a[i - 1], a[i - 1], 0.0)               # ReLU(a_{i-1}) = ReLU(e_{i-1})
            ustats[i] = self._norm_fwd(un, t, View(buf, co, co), RE, 0.0)
            ts[i], U[i - 1] = t, View(buf)
            src = U[i - 1]
        uc0 = lv[0][2]
        o4 = View(new_act(N, H, W, uc0.ipad, dev))
        uc0.fwd(src, o4)                                  # (+ bias)
        pre = torch.empty((N, self.output_nc, H, W), dtype=torch.float32, device=dev)
        ops.nhwc_to_nchw(View(o4.buf, 0, self.output_nc), pre)
        out = torch.empty_like(pre)
        ops.tanh_fwd(pre, out)
        saved = dict(xin=xin, a=a, zs=zs, dstats=dstats, U=U, ts=ts, ustats=ustats, out=out) if save else None
        return out, saved

    # ------------------------------------------------------------------ backward
    def engine_backward(self, sv, gout, need_input_grad, need_param_grad):
        Wg = need_param_grad
        gout = gout.contiguous()
        dev = gout.device
        lv, L = self._lv, self.num_downs
        xin, a, zs, dstats, U, ts, ustats, out = (sv[k] for k in ("xin", "a", "zs", "dstats", "U", "ts", "ustats", "out"))
        N, _, H, W = gout.shape
        gpre = torch.empty_like(gout)
        ops.tanh_bwd(gout, out, gpre)
        uc0 = lv[0][2]
        g4 = View(new_act(N, H, W, uc0.ipad, dev))
        ops.nchw_to_nhwc(gpre, g4, Cpad=uc0.ipad)
        # ---- decoder, outermost inwards.  gU[i]: gradient of U[i] = [ReLU(a_i) | ReLU(d_{i+1})]
        if Wg:
            uc0.wgrad(g4, U[0])
        gU = [None] * L
        gU[0] = View(new_act(N, U[0].H, U[0].W, U[0].C, dev))
        uc0.bwd_data(g4, gU[0])
        g_inner = None                                    # gradient of ReLU(e_innermost)
        for i in range(1, L):
            uc, un = lv[i][2], lv[i][3]
            co = ts[i].C
            gt = View(new_act(N, ts[i].H, ts[i].W, co, dev))
            # through ReLU (gate read from the stored ReLU(norm(t))) and the normalisation
            self._norm_bwd(un, ustats[i], View(gU[i - 1].buf, co, co), View(U[i - 1].buf, co, co), ts[i], gt, 0.0, Wg)
            x_small = U[i] if i < L - 1 else a[L - 1]
            if Wg:
                uc.wgrad(gt, x_small)
            gs = View(new_act(N, x_small.H, x_small.W, x_small.C, dev))
            uc.bwd_data(gt, gs)
            if i < L - 1:
                gU[i] = gs
            else:
                g_inner = gs
        # ---- encoder, innermost outwards.  ga = gradient of a_i through its encoder consumer; the skip adds gU[i][:C] * ReLU'(a_i)
        ops.mask_mul(g_inner, a[L - 1], 0.0)              # ReLU' of the innermost activation -> gradient of its convolution output
        gz = g_inner
        for i in range(L - 1, -1, -1):
            dc, dn = lv[i][0], lv[i][1]
            x_in = a[i - 1] if i > 0 else xin
            if Wg:
                dc.wgrad(x_in, gz)
            if i == 0:
                if not need_input_grad:
                    return None
                gx = View(new_act(N, H, W, xin.C, dev))
                dc.dgrad(gz, gx)
                gin = torch.empty((N, self.input_nc, H, W), dtype=torch.float32, device=dev)
                ops.nhwc_to_nchw(View(gx.buf, 0, self.input_nc), gin)
                return gin
            C = a[i - 1].C
            skip = View(new_act(N, a[i - 1].H, a[i - 1].W, C, dev))
            ops.mask_copy(skip, View(gU[i - 1].buf, 0, C), a[i - 1], 0.0)          # the decoder saw ReLU(a_{i-1})
            ga = View(new_act(N, a[i - 1].H, a[i - 1].W, C, dev))
            dc.dgrad(gz, ga, r1=skip, beta1=1.0)                                    # + the encoder path: conv_i's data-gradient
            if i - 1 == 0:
                ops.mask_mul(ga, a[0], 0.2)               # a_0 = LeakyReLU(conv_0(x)): no normalisation at the outermost level
                gz = ga
            else:
                gzn = View(new_act(N, zs[i - 1].H, zs[i - 1].W, C, dev))
                self._norm_bwd(lv[i - 1][1], dstats[i - 1], ga, a[i - 1], zs[i - 1], gzn, 0.2, Wg)
                gz = gzn
        return None
