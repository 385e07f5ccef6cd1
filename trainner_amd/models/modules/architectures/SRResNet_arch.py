"""SRResNet (BASELINE config 1 generator) on the MI355X engine.

Constructor, state_dict keys and arithmetic follow codes/models/modules/architectures/
SRResNet_arch.py (SRResNet :16-60, ResNetBlock :62-92) as `options/defaults.py:98-112` builds it:
mode CNA, no norm, ReLU, pixel-shuffle upsampling, res_scale 1.  conv+bias+ReLU and the
`x + res` skip are epilogues of the implicit-GEMM kernel.
"""
import math

import torch
import torch.nn as nn

from .... import ops
from ....engine import ConvOp, HipNet
from ....ops import View, new_act
from . import block as B


class ResNetBlock(nn.Module):
    def __init__(self, nf, act_type="relu"):
        super().__init__()
        self.res = B.flat_sequential(B.conv_block(nf, nf, 3, act_type=act_type), B.conv_block(nf, nf, 3, act_type=None))


class SRResNet(HipNet):
    def __init__(self, in_nc, out_nc, nf, nb, upscale=4, norm_type=None, act_type="relu", mode="CNA", res_scale=1,
                 upsample_mode="pixelshuffle", convtype="Conv2D", finalact=None):
        super().__init__()
        if norm_type is not None or mode != "CNA" or convtype != "Conv2D" or finalact or res_scale != 1:
            raise NotImplementedError("SRResNet option outside the default recipe is not implemented by the HIP engine")
        if upsample_mode != "pixelshuffle" or upscale not in (2, 4, 8):
            raise NotImplementedError("upsample mode [%s] is not implemented by the HIP SRResNet" % upsample_mode)
        if nf % 32 or in_nc > 4 or out_nc > 4:
            raise NotImplementedError("HIP SRResNet needs nf %% 32 == 0 and <= 4 image channels")
        self.in_nc, self.out_nc, self.nf, self.nb = in_nc, out_nc, nf, nb
        self.n_up = int(math.log(upscale, 2))
        self.act, self.slope = B.act_code(act_type)
        fea_conv = B.conv_block(in_nc, nf, 3, act_type=None)
        trunk = [ResNetBlock(nf, act_type) for _ in range(nb)] + [B.conv_block(nf, nf, 3, act_type=None)]
        ups = [B.flat_sequential(B.conv_block(nf, nf * 4, 3, act_type=None), B.Marker("pixelshuffle x2"),
                                 B.Marker("act:" + act_type)) for _ in range(self.n_up)]
        hr0 = B.conv_block(nf, nf, 3, act_type=act_type)
        hr1 = B.conv_block(nf, out_nc, 3, act_type=None)
        self.model = B.flat_sequential(fea_conv, B.ShortcutBlock(B.flat_sequential(*trunk)), *ups, hr0, hr1)
        self._init_engine()

    def _build_ops(self, packer):
        m = self.model
        sub = m[1].sub
        o = {"fea": ConvOp(m[0], packer, need_dgrad=False),
             "blk": [(ConvOp(sub[b].res[0], packer), ConvOp(sub[b].res[2], packer)) for b in range(self.nb)],
             "lr": ConvOp(sub[self.nb], packer), "up": []}
        idx = 2
        for _ in range(self.n_up):
            o["up"].append(ConvOp(m[idx], packer))
            idx += 3
        o["hr0"] = ConvOp(m[idx], packer)
        o["hr1"] = ConvOp(m[idx + 2], packer)
        self._ops = o

    def engine_forward(self, x, save):
        o, nf, act, sl = self._ops, self.nf, self.act, self.slope
        N, _, h, w = x.shape
        dev = x.device
        lr = new_act(N, h, w, 4, dev)
        ops.nchw_to_nhwc(x, View(lr), Cpad=4)
        fea = View(new_act(N, h, w, nf, dev))
        o["fea"].fwd(View(lr), fea)
        t, blocks = fea, []
        for c0, c1 in o["blk"]:
            a = View(new_act(N, h, w, nf, dev))
            c0.fwd(t, a, act=act, slope=sl)
            t2 = View(new_act(N, h, w, nf, dev))
            c1.fwd(a, t2, r1=t)                               # x + res(x)
            blocks.append((t, a))
            t = t2
        y0 = View(new_act(N, h, w, nf, dev))
        o["lr"].fwd(t, y0, r1=fea)
        cur, stages = y0, []
        for u in o["up"]:
            nxt = View(new_act(N, cur.H * 2, cur.W * 2, nf, dev))
            if not u.fwd_shuffle2(cur, nxt, act=act, slope=sl):          # the shuffle folded into the convolution's store where the kernel offers it
                tt = View(new_act(N, cur.H, cur.W, 4 * nf, dev))
                u.fwd(cur, tt, act=act, slope=sl)
                ops.depth_to_space(tt, nxt)
            stages.append((cur, nxt))
            cur = nxt
        h0 = View(new_act(N, cur.H, cur.W, nf, dev))
        o["hr0"].fwd(cur, h0, act=act, slope=sl)
        o4 = new_act(N, cur.H, cur.W, 4, dev)
        o["hr1"].fwd(h0, View(o4, 0, self.out_nc))
        out = torch.empty((N, self.out_nc, cur.H, cur.W), dtype=torch.float32, device=dev)
        ops.nhwc_to_nchw(View(o4, 0, self.out_nc), out)
        saved = dict(lr=lr, fea=fea, blocks=blocks, trunk_in=t, stages=stages, hr_in=cur, h0=h0) if save else None
        return out, saved

    def engine_backward(self, sv, gout, need_input_grad, need_param_grad):
        o, nf, sl, W = self._ops, self.nf, self.slope, need_param_grad
        gout = gout.contiguous()
        dev = gout.device
        cur, h0 = sv["hr_in"], sv["h0"]
        N = cur.N
        g4 = new_act(N, cur.H, cur.W, 4, dev)
        ops.nchw_to_nhwc(gout, View(g4), Cpad=4)
        gh0 = View(new_act(N, cur.H, cur.W, nf, dev))
        o["hr1"].dgrad(View(g4), gh0, mask=h0, m_slope=sl)
        if W:
            o["hr1"].wgrad(h0, View(g4, 0, self.out_nc))
        gcur = View(new_act(N, cur.H, cur.W, nf, dev))
        o["hr0"].dgrad(gh0, gcur)
        if W:
            o["hr0"].wgrad(cur, gh0)
        for si in range(len(sv["stages"]) - 1, -1, -1):
            src, dst = sv["stages"][si]
            gt = View(new_act(N, src.H, src.W, 4 * nf, dev))
            ops.space_to_depth_bwd(gcur, gt, mask=dst, mslope=sl)
            if W:
                o["up"][si].wgrad(src, gt)
            gsrc = View(new_act(N, src.H, src.W, nf, dev))
            o["up"][si].dgrad(gt, gsrc)
            gcur = gsrc
        gy0 = gcur
        if W:
            o["lr"].wgrad(sv["trunk_in"], gy0)
        gt_ = View(new_act(N, gy0.H, gy0.W, nf, dev))
        o["lr"].dgrad(gy0, gt_)
        for (tin, a), (c0, c1) in zip(reversed(sv["blocks"]), reversed(o["blk"])):
            ga = View(new_act(N, gy0.H, gy0.W, nf, dev))
            c1.dgrad(gt_, ga, mask=a, m_slope=sl)
            if W:
                c1.wgrad(a, gt_)
                c0.wgrad(tin, ga)
            gin = View(new_act(N, gy0.H, gy0.W, nf, dev))
            c0.dgrad(ga, gin, r1=gt_)                         # skip connection
            gt_ = gin
        ops.axpby(gt_, gy0, 1.0, 1.0)                          # ShortcutBlock
        if W:
            o["fea"].wgrad(View(sv["lr"], 0, self.in_nc), gt_)
        if need_input_grad:
            raise NotImplementedError("gradient w.r.t. the LR input is not on the SR training path")
        return None

    def forward(self, x, outm=None):
        if outm:
            raise NotImplementedError("finalcap/outm [%s] is not implemented by the HIP engine" % outm)
        return super().forward(x)
