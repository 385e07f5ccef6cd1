"""TEST INFRASTRUCTURE ONLY: a torch-CPU stand-in for `trainner_amd.ops`, installed by monkeypatch.

It re-states the *contract* of every C-ABI op (views, epilogues, accumulation semantics) with stock
torch functions so that the HOST logic -- the hand-written forward/backward kernel schedules of the
networks, the autograd bridge, flat parameters, fused clip/Adam plumbing, the SRModel step -- can be
exercised and checked against the reference-generated goldens on a machine without a GPU
(`-m "not gpu"`).  The product never imports this module; on a GPU box the `-m gpu` tests run the
same schedules through libtrainner_hip.so.
"""
import math

import torch
import torch.nn.functional as F

from trainner_amd import hip, ops


class _Packed:
    def __init__(self, w, kind):
        self.w, self.kind = w, kind
        self.KinP = self.KoutP = 0
        self.t = w


class EmulPacker:
    def __init__(self, device):
        self.device = device
        self.jobs = []

    def add(self, w, kind):
        self.jobs.append(_Packed(w, kind))
        return len(self.jobs) - 1

    def get(self, idx):
        return self.jobs[idx]

    def run(self):
        pass


class EmulDensePacker:
    def __init__(self, device):
        self.device = device
        self.jobs = []

    def add_block(self, weights, nf, gc, scale5):
        idx = []
        for t in range(5):
            p = _Packed(None, ops.PACK_DENSE_DGRAD)
            p.srcs, p.tt, p.nf, p.gc, p.scale5 = list(weights), t, nf, gc, scale5
            self.jobs.append(p)
            idx.append(len(self.jobs) - 1)
        return idx

    def get(self, idx):
        return self.jobs[idx]

    def run(self):
        pass


def _dense_dgrad(xin, wp, yC):
    """sum over the consumers of the target group of conv_transpose(g_k, W_k[:, target])."""
    nf, gc, t = wp.nf, wp.gc, wp.tt
    tlo, ntar = (0, nf) if t == 4 else (nf + (3 - t) * gc, gc)
    w5 = _mm(wp.scale5 * wp.srcs[4].detach()[:, tlo:tlo + ntar])       # conv5's scale is folded into the PACKED weight
    out = F.conv_transpose2d(xin[:, :nf], w5, None, padding=1)
    for m in range(t):
        wk = _mm(wp.srcs[3 - m].detach()[:, tlo:tlo + ntar])
        out = out + F.conv_transpose2d(xin[:, nf + m * gc: nf + (m + 1) * gc], wk, None, padding=1)
    return out


def _nchw(v):
    return v.dense().permute(0, 3, 1, 2)


def _fit(t, C):
    """slice / zero-pad channel dim of an NCHW tensor to C channels."""
    if t.shape[1] >= C:
        return t[:, :C]
    return F.pad(t, (0, 0, 0, 0, 0, C - t.shape[1]))


def _mask(m, slope):
    return torch.where(m > 0, torch.ones_like(m), torch.full_like(m, slope))


def _mm(t):
    """TNR_MMA_BF16 contract: operands rounded to bf16 in front of the matrix core (fp32 accumulate)."""
    return t.to(torch.bfloat16).to(torch.float32) if (t is not None and ops.MMA == hip.MMA_BF16) else t


def conv(x, wp, y, mode=ops.CONV_3x3, bias=None, act=ops.ACT_NONE, slope=0.2, alpha=1.0, r1=None, r1_ch=None,
         beta1=1.0, r2=None, alpha2=1.0, mask=None, m_lo=0, m_hi=None, m_slope=0.2, reflect=False, noise=None, wino=None):
    xin = _mm(_nchw(x))      # (wino: a kernel-form choice of the real library; the contract's arithmetic is the same convolution)
    w = None if wp.kind == ops.PACK_DENSE_DGRAD else _mm(wp.w.detach())
    if wp.kind == ops.PACK_DENSE_DGRAD:
        out = _dense_dgrad(xin, wp, y.C)
    elif wp.kind in (ops.PACK_FWD, ops.PACK_C4_FWD):
        xi = _fit(xin, w.shape[1])
        if mode == ops.CONV_3x3_UP2:
            xi = F.interpolate(xi, scale_factor=2.0, mode="nearest")
        kp = w.shape[-1] // 2           # (3x3: 1; the 7x7 image layers of TNR_CONV_7x7_C4: 3)
        out = F.conv2d(F.pad(xi, (kp, kp, kp, kp), mode="reflect"), w, None) if reflect else F.conv2d(xi, w, None, padding=kp)
    elif wp.kind == ops.PACK_FWD_S2D:
        out = F.conv2d(_fit(xin, w.shape[1]), w, None, stride=2, padding=1)
    elif wp.kind in (ops.PACK_DGRAD_3x3, ops.PACK_C4_DGRAD3):
        out = F.conv_transpose2d(_fit(xin, w.shape[0]), w, None, padding=w.shape[-1] // 2)
    else:
        out = F.conv_transpose2d(_fit(xin, w.shape[0]), w, None, stride=2, padding=1)
    _epilogue(out, y, bias, act, slope, alpha, r1, r1_ch, beta1, r2, alpha2, mask, m_lo, m_hi, m_slope, noise)


def _noise_mult(nz, N, H, W, Cc):
    """[N, C, H, W] multiplier field of an ops.Noise (oracle/gauss_noise.py restates csrc/gauss_noise.h)."""
    from oracle import gauss_noise
    m = gauss_noise.multiplier(N * H * W, Cc, nz.sigma, nz.key0, nz.key1, nz.pix0)
    return torch.from_numpy(m).view(N, H, W, Cc).permute(0, 3, 1, 2)


def gauss_mult(dst, src, noise):
    m = _noise_mult(noise, dst.N, dst.H, dst.W, dst.C)
    dst.dense().copy_((m if src is None else _nchw(src) * m).permute(0, 2, 3, 1))


def _epilogue(out, y, bias=None, act=ops.ACT_NONE, slope=0.2, alpha=1.0, r1=None, r1_ch=None, beta1=1.0, r2=None, alpha2=1.0,
              mask=None, m_lo=0, m_hi=None, m_slope=0.2, noise=None):
    out = _fit(out, y.C)
    if bias is not None:
        out = out + _fit(bias.detach().view(1, -1, 1, 1), y.C)
    if act == ops.ACT_LRELU:
        out = F.leaky_relu(out, slope)
    elif act == ops.ACT_RELU:
        out = F.relu(out)
    out = out * alpha
    if r1 is not None:
        ch = r1.C if r1_ch is None else r1_ch
        out = torch.cat([out[:, :ch] + beta1 * _nchw(r1)[:, :ch], out[:, ch:]], 1)
    if noise is not None and noise.pos == 1:
        out = out * _noise_mult(noise, y.N, y.H, y.W, y.C)
    if r2 is not None:
        out = out * alpha2 + _nchw(r2)[:, :y.C]
    if noise is not None and noise.pos == 2:
        out = out * _noise_mult(noise, y.N, y.H, y.W, y.C)
    if mask is not None:
        hi = y.C if m_hi is None else m_hi
        mm = _mask(_nchw(mask)[:, m_lo:hi], m_slope)
        out = torch.cat([out[:, :m_lo], out[:, m_lo:hi] * mm, out[:, hi:]], 1)
    y.dense().copy_(out.permute(0, 2, 3, 1))


def conv_col(x, wp, y, k, stride=1, pad=1, **epi):
    xin, w = _mm(_nchw(x)), _mm(wp.w.detach())
    if wp.kind == ops.PACK_COL_FWD:
        out = F.conv2d(_fit(xin, w.shape[1]), w, None, stride=stride, padding=pad)
    else:       # PACK_COL_DGRAD3: x is the gradient of the layer's output, pad = k - 1 - (the layer's padding)
        out = F.conv_transpose2d(_fit(xin, w.shape[0]), w, None, stride=1, padding=k - 1 - pad)
    _epilogue(out, y, **epi)


def window2d(src, dst, oy, ox, acc=False):
    s = src.dense()
    d = dst.dense()
    if not acc:
        d.zero_()
    y0, x0 = max(0, -oy), max(0, -ox)
    y1, x1 = min(dst.H, src.H - oy), min(dst.W, src.W - ox)
    if y1 > y0 and x1 > x0:
        d[:, y0:y1, x0:x1].add_(s[:, y0 + oy:y1 + oy, x0 + ox:x1 + ox])


def conv_thin(x, w, y, bias=None, alpha=1.0, dgrad=False):
    xin, wt = _nchw(x), w.detach()
    out = F.conv_transpose2d(_fit(xin, wt.shape[0]), wt, None, padding=1) if dgrad else F.conv2d(_fit(xin, wt.shape[1]), wt, None, padding=1)
    if bias is not None:
        out = out + bias.detach().view(1, -1, 1, 1)
    ops.View(y.buf, y.coff, out.shape[1]).dense().copy_((out * alpha).permute(0, 2, 3, 1))


def wgrad_thin(big, small4, dw, db, flip, alpha=1.0, beta=1.0):
    """flip False: layer small(<= 3 ch) -> big: x = small4, g = big;  flip True: layer big -> small: x = big, g = small4."""
    xv, gv = (big, small4) if flip else (small4, big)
    O, I = dw.shape[0], dw.shape[1]
    xin, gin = _nchw(xv)[:, :I], _nchw(gv)[:, :O]
    with torch.enable_grad():
        w0 = torch.zeros(O, I, 3, 3, requires_grad=True)
        (gw,) = torch.autograd.grad(F.conv2d(xin, w0, None, padding=1), w0, gin)
    dw.copy_(beta * dw + alpha * gw)
    if db is not None:
        db.copy_(beta * db + alpha * gin.sum(dim=(0, 2, 3)))


def conv_thin7(x, w, y, pad=3, reflect=True, bias=None, alpha=1.0, dgrad=False):
    xin, wt = _nchw(x), w.detach()
    if dgrad:       # out[q] = sum_t W^T[t] x[q + (6 - t) - pad]: the full correlation for pad = 6
        out = F.conv_transpose2d(_fit(xin, wt.shape[0]), wt, None, padding=6 - pad)
    else:
        xi = _fit(xin, wt.shape[1])
        out = F.conv2d(F.pad(xi, (pad,) * 4, mode="reflect"), wt, None) if reflect else F.conv2d(xi, wt, None, padding=pad)
    assert out.shape[2] == y.H and out.shape[3] == y.W, (out.shape, y.H, y.W)
    if bias is not None:
        out = out + bias.detach().view(1, -1, 1, 1)
    ops.View(y.buf, y.coff, out.shape[1]).dense().copy_((out * alpha).permute(0, 2, 3, 1))


def wgrad_thin7(big, small4, dw, db, flip, rpad=0, off=0, alpha=1.0, beta=1.0):
    """flip False: layer image -> big over the reflection-padded image small4;  flip True: layer big -> image, big read through
    ReflectionPad2d(rpad), small4 = gradient of the output."""
    O, I = dw.shape[0], dw.shape[1]
    if flip:
        assert rpad == 3 and off == -6 and db is None
        xin, gin = F.pad(_nchw(big)[:, :I], (3, 3, 3, 3), mode="reflect"), _nchw(small4)[:, :O]
    else:
        assert rpad == 0 and off == 0
        xin, gin = _nchw(small4)[:, :I], _nchw(big)[:, :O]
    with torch.enable_grad():
        w0 = torch.zeros(O, I, 7, 7, requires_grad=True)
        (gw,) = torch.autograd.grad(F.conv2d(xin, w0, None), w0, gin)
    dw.copy_(beta * dw + alpha * gw)
    if db is not None:
        db.copy_(beta * db + alpha * gin.sum(dim=(0, 2, 3)))


def wgrad(x, g, dw, db=None, mode=ops.CONV_3x3, cin_begin=0, alpha=1.0, beta=1.0, reflect=False):
    k = 4 if mode == ops.CONV_4x4_S2 else 3
    xin, gin_raw = _mm(_nchw(x)), _nchw(g)
    gin = _mm(gin_raw)
    with torch.enable_grad():
        w0 = torch.zeros(g.C, x.C, k, k, requires_grad=True)
        if mode == ops.CONV_3x3_UP2:
            out = F.conv2d(F.interpolate(xin, scale_factor=2.0, mode="nearest"), w0, None, padding=1)
        elif mode == ops.CONV_4x4_S2:
            out = F.conv2d(xin, w0, None, stride=2, padding=1)
        elif reflect:
            out = F.conv2d(F.pad(xin, (1, 1, 1, 1), mode="reflect"), w0, None)
        else:
            out = F.conv2d(xin, w0, None, padding=1)
        (gw,) = torch.autograd.grad(out, w0, gin)
    tgt = dw[:, cin_begin:cin_begin + x.C]
    tgt.copy_(beta * tgt + alpha * gw)
    if db is not None:
        db.copy_(beta * db + alpha * gin_raw.sum(dim=(0, 2, 3)))


def nchw_to_nhwc(src, dst, Cpad=None, scale=None, shift=None):
    t = src
    if scale is not None:
        t = t * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    Cpad = dst.C if Cpad is None else Cpad
    View = ops.View
    View(dst.buf, dst.coff, Cpad).dense().copy_(_fit(t, Cpad).permute(0, 2, 3, 1))


def nhwc_to_nchw(src, dst, scale=None, accumulate=False):
    t = _nchw(src)[:, :dst.shape[1]]
    if scale is not None:
        t = t * scale.view(1, -1, 1, 1)
    if accumulate:
        dst.add_(t)
    else:
        dst.copy_(t)


def upsample2x_bwd(gup, gx, mask=None, mslope=0.2):
    t = F.avg_pool2d(_nchw(gup), 2) * 4
    if mask is not None:
        t = t * _mask(_nchw(mask), mslope)
    gx.dense().copy_(t.permute(0, 2, 3, 1))


def depth_to_space(x, y):
    y.dense().copy_(F.pixel_shuffle(_nchw(x), 2).permute(0, 2, 3, 1))


def space_to_depth_bwd(gy, gx, mask=None, mslope=0.2):
    t = _nchw(gy)
    if mask is not None:
        t = t * _mask(_nchw(mask), mslope)
    gx.dense().copy_(F.pixel_unshuffle(t, 2).permute(0, 2, 3, 1))


def maxpool2_fwd(x, y):
    y.dense().copy_(F.max_pool2d(_nchw(x), 2, 2).permute(0, 2, 3, 1))


def maxpool2_bwd(gy, x, gx):
    with torch.enable_grad():
        xin = _nchw(x).clone().requires_grad_(True)
        (g,) = torch.autograd.grad(F.max_pool2d(xin, 2, 2), xin, _nchw(gy))
    gx.dense().copy_((g * (xin.detach() > 0)).permute(0, 2, 3, 1))


def bilinear2x_fwd(x, y):
    y.dense().copy_(F.interpolate(_nchw(x), scale_factor=2, mode="bilinear", align_corners=False).permute(0, 2, 3, 1))


def bilinear2x_bwd(gy, gx=None, gz=None, mask=None, mslope=0.2):
    ref = gx if gx is not None else gz
    with torch.enable_grad():
        xin = torch.zeros(ref.N, ref.C, ref.H, ref.W, requires_grad=True)
        (g,) = torch.autograd.grad(F.interpolate(xin, scale_factor=2, mode="bilinear", align_corners=False), xin, _nchw(gy))
    if gx is not None:
        gx.dense().copy_(g.permute(0, 2, 3, 1))
    if gz is not None:
        gz.dense().copy_((g * _mask(_nchw(mask), mslope)).permute(0, 2, 3, 1))


def add2(dst, a, b):
    dst.dense().copy_(a.dense() + b.dense())


def mask_copy(dst, src, y, mslope=0.2):
    dst.dense().copy_(src.dense() * _mask(y.dense(), mslope))


def _gpad(t, pad, reflect):
    return F.pad(t, (pad,) * 4, mode="reflect") if (reflect and pad) else F.pad(t, (pad,) * 4)


def gconv_fwd(x, w, y, bias=None, stride=1, pad=0, reflect=False, act=ops.ACT_NONE, slope=0.2):
    o = F.conv2d(_gpad(_nchw(x)[:, :w.shape[1]], pad, reflect), w.detach(), None if bias is None else bias.detach(), stride=stride)
    o = F.leaky_relu(o, slope) if act == ops.ACT_LRELU else (F.relu(o) if act == ops.ACT_RELU else o)
    y.dense().copy_(_fit(o, y.C).permute(0, 2, 3, 1))


def gconv_dgrad(g, w, gx, stride=1, pad=0, reflect=False):
    with torch.enable_grad():
        xin = torch.zeros(gx.N, w.shape[1], gx.H, gx.W, requires_grad=True)
        (gr,) = torch.autograd.grad(F.conv2d(_gpad(xin, pad, reflect), w.detach(), None, stride=stride), xin, _nchw(g)[:, :w.shape[0]])
    gx.dense().copy_(_fit(gr, gx.C).permute(0, 2, 3, 1))


def gconv_wgrad(x, g, dw, db=None, stride=1, pad=0, reflect=False, alpha=1.0, beta=1.0):
    gin = _nchw(g)[:, :dw.shape[0]]
    with torch.enable_grad():
        w0 = torch.zeros_like(dw, requires_grad=True)
        (gw,) = torch.autograd.grad(F.conv2d(_gpad(_nchw(x)[:, :dw.shape[1]], pad, reflect), w0, None, stride=stride), w0, gin)
    dw.copy_(beta * dw + alpha * gw)
    if db is not None:
        db.copy_(beta * db + alpha * gin.sum(dim=(0, 2, 3)))


def bias_grad(g, db, alpha=1.0, beta=1.0):
    db.copy_(beta * db + alpha * g.dense().sum(dim=(0, 1, 2)))


def pad2d(x, y, pad, reflect):
    y.dense().copy_(_gpad(_nchw(x), pad, reflect).permute(0, 2, 3, 1))


def unpad2d(xp, y, pad, fold):
    t = _nchw(xp)
    if not fold:
        y.dense().copy_(t[:, :, pad:pad + y.H, pad:pad + y.W].permute(0, 2, 3, 1))
        return
    with torch.enable_grad():
        xin = torch.zeros(y.N, y.C, y.H, y.W, requires_grad=True)
        (gr,) = torch.autograd.grad(_gpad(xin, pad, True), xin, t)
    y.dense().copy_(gr.permute(0, 2, 3, 1))


def tanh_fwd(x, y):
    y.copy_(torch.tanh(x))


def tanh_bwd(g, y, gx):
    gx.copy_(g * (1 - y * y))


def gan_loss(pred, kind, target, out, grad=None):
    p = pred.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        t = torch.full_like(p, float(target))
        l = F.binary_cross_entropy_with_logits(p, t) if kind == 0 else (F.mse_loss(p, t) if kind == 1 else p.mean())
        (g,) = torch.autograd.grad(l, p)
    out.reshape(-1)[0] = l.detach()
    if grad is not None:
        grad.copy_(g)


def axpby(dst, src, a=1.0, b=1.0):
    d = dst.dense()
    d.copy_(a * src.dense() + (b * d if b != 0 else 0))


def mask_mul(g, y, mslope=0.2):
    g.dense().mul_(_mask(y.dense(), mslope))


def fill(t, value=0.0):
    t.fill_(value)


def bn_replay_running(running_mean, running_var, num_batches, stat64, momentum=0.1):
    C = running_mean.numel()
    running_mean.copy_(((1.0 - momentum) * running_mean.double() + momentum * stat64[:C]).float())
    running_var.copy_(((1.0 - momentum) * running_var.double() + momentum * stat64[C:]).float())
    if num_batches is not None:
        num_batches += 1


def bn_train_fwd(z, y, gamma, beta, running_mean, running_var, num_batches, save_mean, save_invstd, momentum=0.1,
                 eps=1e-5, act=ops.ACT_LRELU, slope=0.2, stat64=None):
    zin = _nchw(z)
    C = zin.shape[1]
    n = zin.numel() / C
    mean64 = zin.double().mean(dim=(0, 2, 3))
    unb64 = zin.double().var(dim=(0, 2, 3), unbiased=False) * (n / max(n - 1, 1))
    if stat64 is not None:
        stat64[:C].copy_(mean64)
        stat64[C:].copy_(unb64)
    if running_mean is not None:          # the kernel's update: fp64 arithmetic, one rounding to fp32 (tnr_bn_replay_running repeats it)
        running_mean.copy_(((1.0 - momentum) * running_mean.double() + momentum * mean64).float())
        running_var.copy_(((1.0 - momentum) * running_var.double() + momentum * unb64).float())
    mean = zin.mean(dim=(0, 2, 3))
    var = zin.var(dim=(0, 2, 3), unbiased=False)
    save_mean.copy_(mean)
    save_invstd.copy_(1.0 / torch.sqrt(var + eps))
    out = F.batch_norm(zin, None, None, gamma.detach(), beta.detach(), True, momentum, eps)
    if num_batches is not None:
        num_batches += 1
    out = F.leaky_relu(out, slope) if act == ops.ACT_LRELU else (F.relu(out) if act == ops.ACT_RELU else out)
    y.dense().copy_(out.permute(0, 2, 3, 1))


def bn_train_bwd(gy, y, z, gz, gamma, save_mean, save_invstd, dgamma=None, dbeta=None, acc_beta=1.0, mslope=0.2, beta=None):
    g = _nchw(gy) * _mask(_nchw(y), mslope)
    zin = _nchw(z)
    n = zin.numel() / zin.shape[1]
    xm = zin - save_mean.view(1, -1, 1, 1)
    s, dot = g.sum(dim=(0, 2, 3)), (g * xm).sum(dim=(0, 2, 3))
    inv = save_invstd
    k = (dot * inv * inv / n).view(1, -1, 1, 1)
    dz = (g - (s / n).view(1, -1, 1, 1) - xm * k) * (inv * gamma.detach()).view(1, -1, 1, 1)
    gz.dense().copy_(dz.permute(0, 2, 3, 1))
    if dgamma is not None:
        dgamma.copy_(acc_beta * dgamma + dot * inv)
        dbeta.copy_(acc_beta * dbeta + s)


def instnorm_fwd(z, y, save_mean, save_invstd, eps=1e-5, act=ops.ACT_NONE, slope=0.0):
    zin = _nchw(z)
    mean = zin.mean(dim=(2, 3))
    var = zin.var(dim=(2, 3), unbiased=False)
    save_mean.copy_(mean.reshape(-1))
    save_invstd.copy_((1.0 / torch.sqrt(var + eps)).reshape(-1))
    out = (zin - mean[:, :, None, None]) * (1.0 / torch.sqrt(var + eps))[:, :, None, None]
    out = F.leaky_relu(out, slope) if act == ops.ACT_LRELU else (F.relu(out) if act == ops.ACT_RELU else out)
    y.dense().copy_(out.permute(0, 2, 3, 1))


def instnorm_bwd(gy, y, z, gz, save_mean, save_invstd, mslope=1.0):
    g = _nchw(gy) * _mask(_nchw(y), mslope)
    zin = _nchw(z)
    N, C = zin.shape[0], zin.shape[1]
    n = zin.shape[2] * zin.shape[3]
    mean, inv = save_mean.view(N, C, 1, 1), save_invstd.view(N, C, 1, 1)
    xm = zin - mean
    s, dot = g.sum(dim=(2, 3), keepdim=True), (g * xm).sum(dim=(2, 3), keepdim=True)
    dz = (g - s / n - xm * (dot * inv * inv / n)) * inv
    gz.dense().copy_(dz.permute(0, 2, 3, 1))


def linear_fwd(x, w, b, y, act=ops.ACT_NONE, slope=0.2):
    o = F.linear(x, w.detach(), None if b is None else b.detach())
    y.copy_(F.leaky_relu(o, slope) if act == ops.ACT_LRELU else o)


def linear_bwd(x, w, gy, yact=None, gx=None, dw=None, db=None, acc_beta=1.0, mslope=0.2):
    gpre = gy if yact is None else gy * _mask(yact, mslope)
    if dw is not None:
        dw.copy_(acc_beta * dw + gpre.t() @ x)
    if db is not None:
        db.copy_(acc_beta * db + gpre.sum(0))
    if gx is not None:
        gx.copy_(gpre @ w.detach())


def l1_mean_fwd(a, b, scale, out):
    out.copy_(scale * (a.detach() - b).abs().mean())


def l1_mean_bwd(a, b, scale, gscale, ga, accumulate=False):
    g = torch.sign(a.detach() - b) * (scale * (1.0 if gscale is None else float(gscale.reshape(-1)[0])) / a.numel())
    if accumulate:
        ga.add_(g)
    else:
        ga.copy_(g)


def ragan_phase_a(pf, pr, sums):
    sums.zero_()
    sums[0], sums[1], sums[2] = pf.sum(), pr.sum(), float(pf.numel())


def ragan_phase_b(pf, pr, stage, sums):
    mf, mr = sums[0] / sums[2], sums[1] / sums[2]
    dr, df = pr - mf, pf - mr
    if stage == 0:
        sums[3], sums[4], sums[5], sums[6] = F.softplus(dr).sum(), F.softplus(-df).sum(), torch.sigmoid(dr).sum(), 0.0
    else:
        sums[3], sums[4] = F.softplus(-dr).sum(), F.softplus(df).sum()
        sums[5], sums[6] = torch.sigmoid(-dr).sum(), torch.sigmoid(df).sum()


def ragan_phase_c(pf, pr, stage, weight, sums, out, gf, gr):
    NN = sums[2]
    mf, mr = sums[0] / NN, sums[1] / NN
    l1, l2 = sums[3] / NN, sums[4] / NN
    out[0], out[1], out[2], out[3], out[4] = weight * (l1 + l2) * 0.5, l1, l2, mr, mf
    k = weight * 0.5 / NN
    dr, df = pr - mf, pf - mr
    if stage == 0:
        gf.copy_(k * (-(sums[5] / NN) - torch.sigmoid(-df)))
        gr.zero_()
    else:
        gf.copy_(k * (torch.sigmoid(df) + sums[5] / NN))
        gr.copy_(k * (-torch.sigmoid(-dr) - sums[6] / NN))


def scale_by(dst, src, gscale):
    dst.copy_(src * gscale.reshape(-1)[0])


def sumsq(g, out):
    out.copy_((g.double() ** 2).sum().reshape(1))


def clip_by_norm(g, sumsq_t, max_norm):
    coef = min(1.0, max_norm / (math.sqrt(float(sumsq_t[0])) + 1e-6))
    g.mul_(coef)


def adam_step(p, g, m, v, step_size, b1, b2, bc2_sqrt, eps, wd=0.0):
    gr = g + wd * p if wd else g
    m.lerp_(gr, 1 - b1)
    v.mul_(b2).addcmul_(gr, gr, value=1 - b2)
    p.addcdiv_(m, v.sqrt() / bc2_sqrt + eps, value=-step_size)


def conv_chain(stages):
    for st in stages:
        conv(**{k: v for k, v in st.items() if k != "fresh_from"})


def wgrad_group(items, mode=ops.CONV_3x3):
    for it in items:
        kw = dict(mode=mode, cin_begin=it.get("cin_begin", 0), alpha=it.get("alpha", 1.0), beta=it.get("beta", 1.0), reflect=it.get("reflect", False))
        if it.get("pair") is not None:           # cout pair (tnr_wgrad_desc.cout_split): two layers' gradients side by side in g
            dw2, db2, split = it["pair"]
            g = it["g"]
            wgrad(it["x"], ops.View(g.buf, g.coff, split), it["dw"], it.get("db"), **kw)
            wgrad(it["x"], ops.View(g.buf, g.coff + split, g.C - split), dw2, db2, **kw)
        else:
            wgrad(it["x"], it["g"], it["dw"], it.get("db"), **kw)


_NAMES = ["gauss_mult", "bn_replay_running", "instnorm_fwd", "instnorm_bwd", "conv_col", "window2d", "conv_thin", "wgrad_thin", "conv_thin7", "wgrad_thin7", "bias_grad", "gconv_fwd", "gconv_dgrad", "gconv_wgrad", "pad2d", "unpad2d", "tanh_fwd", "tanh_bwd", "gan_loss", "bilinear2x_fwd", "bilinear2x_bwd", "add2", "mask_copy", "conv", "conv_chain", "wgrad", "wgrad_group", "nchw_to_nhwc", "nhwc_to_nchw", "upsample2x_bwd", "depth_to_space", "space_to_depth_bwd",
          "maxpool2_fwd", "maxpool2_bwd", "axpby", "mask_mul", "fill", "bn_train_fwd", "bn_train_bwd", "linear_fwd",
          "linear_bwd", "l1_mean_fwd", "l1_mean_bwd", "ragan_phase_a", "ragan_phase_b", "ragan_phase_c", "scale_by",
          "sumsq", "clip_by_norm", "adam_step"]


class DirectPatcher:
    """monkeypatch stand-in for worker processes (no pytest fixture there; the process exits afterwards)."""

    @staticmethod
    def setattr(obj, name, value):
        setattr(obj, name, value)


def chunked_bn(chunks):
    """(bn_train_fwd, bn_train_bwd) that take BatchNorm statistics per contiguous chunk of the batch: ONE process
    reproducing what `chunks` data-parallel replicas do (per-replica statistics; replica 0's running stats are the
    ones kept: SURVEY.md 8(e))."""
    View = ops.View

    def _split(v, i):
        n = v.N // chunks
        return View(v.buf[i * n:(i + 1) * n], v.coff, v.C)

    stats = {}      # the network allocates [C] statistics; the per-chunk ones live here, keyed by that tensor

    def fwd(z, y, gamma, beta, running_mean, running_var, num_batches, save_mean, save_invstd, **kw):
        per = []
        for i in range(chunks):
            keep = i == 0
            m, s = torch.empty_like(save_mean), torch.empty_like(save_invstd)
            rm = running_mean if keep else (None if running_mean is None else running_mean.clone())
            rv = running_var if keep else (None if running_var is None else running_var.clone())
            kw_i = kw if keep else {k: v for k, v in kw.items() if k != "stat64"}      # (replica 0's statistics are the ones kept)
            bn_train_fwd(_split(z, i), _split(y, i), gamma, beta, rm, rv, num_batches if keep else None, m, s, **kw_i)
            per.append((m, s))
        stats[save_mean.data_ptr()] = (save_mean, per)      # holding save_mean keeps the key unique

    def bwd(gy, y, z, gz, gamma, save_mean, save_invstd, dgamma=None, dbeta=None, acc_beta=1.0, **kw):
        _, per = stats[save_mean.data_ptr()]                 # (kept: a memoized forward's activations serve two backward passes)
        for i in range(chunks):
            bn_train_bwd(_split(gy, i), _split(y, i), _split(z, i), _split(gz, i), gamma, per[i][0], per[i][1],
                         dgamma=dgamma, dbeta=dbeta, acc_beta=acc_beta if i == 0 else 1.0, **kw)

    return fwd, bwd


def install(monkeypatch=None):
    monkeypatch = monkeypatch or DirectPatcher
    g = globals()
    for n in _NAMES:
        monkeypatch.setattr(ops, n, g[n])
    monkeypatch.setattr(ops, "IMAGE_C4", False)
    monkeypatch.setattr(ops, "SMALL_GEMM", False)     # the emulation has no im2col path: layers stay direct convolutions
    monkeypatch.setattr(ops, "WeightPacker", EmulPacker)
    monkeypatch.setattr(ops, "DensePacker", EmulDensePacker)
    monkeypatch.setattr(hip, "require_device", lambda t=None: None)
    monkeypatch.setattr(hip, "engine_device", lambda index=0: torch.device("cpu"))
