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
        cur, acts = x4, []
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
                ops.bn_train_fwd(z, y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                 mean, invstd, momentum=bn.momentum, eps=bn.eps, act=act, slope=sl)
                acts.append((cur, z, y, mean, invstd))
            cur = y
        feat = torch.empty((N, self.final_nc, cur.H, cur.W), dtype=torch.float32, device=dev)   # NCHW flatten order
        ops.nhwc_to_nchw(cur, feat)
        l0, l1 = self.classifier[0], self.classifier[2]
        hid = torch.empty((N, l0.out_features), dtype=torch.float32, device=dev)
        ops.linear_fwd(feat.view(N, -1), l0.weight, l0.bias, hid, act=ops.ACT_LRELU, slope=0.2)
        out = torch.empty((N, 1), dtype=torch.float32, device=dev)
        ops.linear_fwd(hid, l1.weight, l1.bias, out)
        saved = dict(acts=acts, feat=feat, hid=hid, last=cur) if save else None
        return out, saved

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
                                 dbeta=bn.bias.grad if W else None, mslope=sl)
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
