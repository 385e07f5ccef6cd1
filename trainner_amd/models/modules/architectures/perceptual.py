"""VGG feature extractor for the perceptual loss, on the MI355X engine.

Follows codes/models/modules/architectures/perceptual.py FeatureExtractor (:73-214): ImageNet
(x-mean)/std, torchvision cfg-E/D `features` truncated after the last listened layer, listened
features taken BEFORE the ReLU when a 'convX_Y' name is listened (the `x.clone()` at :211-212
happens before the in-place ReLU that follows).  Parameters are frozen (requires_grad False,
:185-188): only the data-gradient schedule exists.

torchvision is not a dependency of the kernels: the layer table is the public VGG configuration.
Weights: `load_path` (a torchvision `vggNN` state_dict, keys `features.N.*`) when given, else
torchvision's pretrained ImageNet weights exactly like the reference (:139-144).  If neither is
available the constructor RAISES -- a perceptual loss against random features is never entered
silently; benchmarks and parity tests (no network access: they load their own seeded weights right
after construction) opt in with `allow_random_init=True` (option `train.perceptual_allow_random_init`).
"""
import logging
import os

import torch
import torch.nn as nn

from .... import ops
from ....engine import ConvOp, HipNet
from ....ops import View, new_act
from . import block as B

logger = logging.getLogger("base")

VGG_CFG = {
    "vgg16": [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"],
    "vgg19": [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"],
}


def vgg_layer_names(net):
    names, blk, idx = [], 1, 1
    for v in VGG_CFG[net]:
        if v == "M":
            names.append("pool%d" % blk)
            blk, idx = blk + 1, 1
        else:
            names += ["conv%d_%d" % (blk, idx), "relu%d_%d" % (blk, idx)]
            idx += 1
    return names


# file names of torchvision's ImageNet checkpoints (torchvision/models/vgg.py model_urls), as cached under <torch hub dir>/checkpoints
TORCHVISION_FILES = {"vgg11": "vgg11-8a719046.pth", "vgg13": "vgg13-19584684.pth", "vgg16": "vgg16-397923af.pth", "vgg19": "vgg19-dcbb9e9d.pth"}


class FeatureExtractor(HipNet):
    def __init__(self, listen_list=None, net="vgg19", use_input_norm=True, z_norm=False, requires_grad=False,
                 remove_pooling=False, pooling_stride=2, change_padding=False, load_path=None, allow_random_init=False):
        super().__init__()
        if net not in VGG_CFG or remove_pooling or pooling_stride != 2 or change_padding or requires_grad or z_norm:
            raise NotImplementedError("FeatureExtractor option outside the ESRGAN recipe is not implemented by the HIP engine")
        listen_list = list(listen_list or ["conv5_4"])
        if len(listen_list) != 1 or not listen_list[0].startswith("conv"):
            raise NotImplementedError("the HIP FeatureExtractor listens to exactly one pre-ReLU conv layer")
        self.listen = listen_list[0]
        self.listen_list = set(listen_list)
        self.use_input_norm = use_input_norm
        names = vgg_layer_names(net)
        last = names.index(self.listen)
        self.names = names[:last + 1]
        layers, c, chans = nn.ModuleDict(), 3, iter([v for v in VGG_CFG[net] if v != "M"])
        for n in self.names:
            if n.startswith("conv"):
                v = next(chans)
                layers[n] = B.Conv2dHIP(c, v, 3, 1)
                c = v
            else:
                layers[n] = B.Marker(n)
        self.feature_net = layers            # keys feature_net.convX_Y.{weight,bias} as in the reference
        self.out_channels = c
        if use_input_norm:
            self.register_buffer("mean", torch.tensor([[[0.485]], [[0.456]], [[0.406]]]))
            self.register_buffer("std", torch.tensor([[[0.229]], [[0.224]], [[0.225]]]))
        self.weights_source = self._load_pretrained(net, load_path, allow_random_init)
        for p in self.parameters():
            p.requires_grad = False
        self.eval()
        self._init_engine()
        self._norm = None

    def _load_pretrained(self, net, load_path, allow_random_init):
        """perceptual.py:134-144: a local torchvision state_dict, else torchvision's ImageNet weights; never a
        silent random init."""
        if load_path and os.path.exists(load_path):
            self.load_torchvision_state(torch.load(load_path, map_location="cpu", weights_only=False))
            return load_path
        why = "pretrained_path %r does not exist" % load_path if load_path else "no perceptual_opt.pretrained_path given"
        try:
            from torchvision.models import vgg as tv_vgg          # optional: not installed on the build image
            self.load_torchvision_state(getattr(tv_vgg, net)(pretrained=True).state_dict())
            return "torchvision:%s" % net
        except Exception as e:                                    # no torchvision / no network / no cached weights
            why += "; torchvision pretrained weights unavailable (%s: %s)" % (type(e).__name__, e)
        # the file torchvision's `pretrained=True` would have cached (torch.hub checkpoints dir, $TORCH_HOME): usable without
        # torchvision itself -- an offline box that carries the cached weights runs the reference's recipe unmodified
        cached = os.path.join(torch.hub.get_dir(), "checkpoints", TORCHVISION_FILES.get(net, ""))
        if os.path.isfile(cached):
            self.load_torchvision_state(torch.load(cached, map_location="cpu", weights_only=False))
            return cached
        why += "; no cached %s" % cached
        if not allow_random_init:
            from ....hip import HipEngineError
            raise HipEngineError("FeatureExtractor(%s): %s. A perceptual loss on randomly initialised features is refused; "
                                 "set train.perceptual_opt.pretrained_path to a torchvision %s state_dict, or opt in with "
                                 "train.perceptual_allow_random_init: true (benchmarks / parity tests that load their own "
                                 "weights)" % (net, why, net))
        logger.warning("FeatureExtractor(%s): %s -- weights left at their random init (allow_random_init)", net, why)
        return "random-init"

    def load_torchvision_state(self, sd):
        """Accept a torchvision vggNN state_dict (features.<i>.weight/bias): torchvision's `features`
        indices run over the same conv / relu / pool sequence as self.names."""
        own = self.state_dict()
        for i, n in enumerate(self.names):
            if n.startswith("conv"):
                own["feature_net.%s.weight" % n].copy_(sd["features.%d.weight" % i])
                own["feature_net.%s.bias" % n].copy_(sd["features.%d.bias" % i])

    def _build_ops(self, packer):
        self._ops = {n: ConvOp(self.feature_net[n], packer) for n in self.names if n.startswith("conv")}

    def _norm_consts(self, dev):
        """(x - mean)/std as x*scale + shift; six host-side constants uploaded once."""
        if self._norm is None or self._norm[0].device != dev:
            if self.use_input_norm:
                mean = [float(v) for v in self.mean.reshape(3).cpu().tolist()]
                std = [float(v) for v in self.std.reshape(3).cpu().tolist()]
                scale = torch.tensor([1.0 / s for s in std], dtype=torch.float32).to(dev)
                shift = torch.tensor([-m / s for m, s in zip(mean, std)], dtype=torch.float32).to(dev)
            else:
                scale = torch.ones(3).to(dev)
                shift = torch.zeros(3).to(dev)
            self._norm = (scale, shift)
        return self._norm

    def engine_forward(self, x, save):
        N, Cc, H, W = x.shape
        if Cc != 3:
            raise ValueError("FeatureExtractor expects RGB input")
        dev = x.device
        scale, shift = self._norm_consts(dev)
        x4 = View(new_act(N, H, W, 4, dev))
        ops.nchw_to_nhwc(x, x4, Cpad=4, scale=scale, shift=shift)
        cur, tape = x4, []
        for n in self.names:
            if n.startswith("conv"):
                y = View(new_act(N, cur.H, cur.W, self.feature_net[n].out_channels, dev))
                if n == self.listen:
                    self._ops[n].fwd(cur, y)                              # listened pre-ReLU
                else:
                    self._ops[n].fwd(cur, y, act=ops.ACT_RELU)
                tape.append((n, cur, y))
                cur = y
            elif n.startswith("pool"):
                y = View(new_act(N, cur.H // 2, cur.W // 2, cur.C, dev))
                ops.maxpool2_fwd(cur, y)
                tape.append((n, cur, y))
                cur = y
        # logical NCHW tensor over the NHWC storage (channels_last strides); L1 is layout agnostic
        out = cur.buf.permute(0, 3, 1, 2)
        return out, (dict(tape=tape, in_shape=(N, H, W)) if save else None)

    def engine_backward(self, sv, gout, need_input_grad, need_param_grad):
        if not need_input_grad:
            return None
        tape = sv["tape"]
        N, H, W = sv["in_shape"]
        dev = gout.device
        g_nhwc = gout.permute(0, 2, 3, 1)
        if not g_nhwc.is_contiguous():
            g_nhwc = g_nhwc.contiguous()
        g = View(g_nhwc)
        for i in range(len(tape) - 1, -1, -1):
            n, xin, y = tape[i]
            gx = View(new_act(N, xin.H, xin.W, xin.C, dev))
            if n.startswith("pool"):
                ops.maxpool2_bwd(g, xin, gx)                              # routes + ReLU' of the pooled activation
            else:
                prev = tape[i - 1][0] if i > 0 else None
                if prev is not None and prev.startswith("conv"):
                    self._ops[n].dgrad(g, gx, mask=xin, m_slope=0.0)      # ReLU' of the producing conv
                else:
                    self._ops[n].dgrad(g, gx)                             # input image / pooled map: no activation
            g = gx
        scale, _ = self._norm_consts(dev)
        out = torch.empty((N, 3, H, W), dtype=torch.float32, device=dev)
        ops.nhwc_to_nchw(View(g.buf, 0, 3), out, scale=scale)
        return out

    def forward(self, x):
        feat = super().forward(x)
        return {self.listen: feat}
