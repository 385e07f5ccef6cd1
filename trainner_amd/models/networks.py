"""Network factory of the MI355X engine -- the reference's operator/plugin API for nets.

Same entry points and option handling as codes/models/networks.py: `define_G(opt, step, net_name)`
(:267), `define_D(opt, net_name)` (:283), `define_F(opt)` (:316), built through `get_network`
(:107-255: pops init_type / init_scale / strict / type, instantiates `cls(**rest)`, applies
`init_weights` when training).  Differences, all deliberate:
  * the classes are the HIP-engine networks (same constructors, same state_dict keys);
  * no nn.DataParallel wrapper (:252-255): data parallelism is one process per GPU with RCCL
    gradient all-reduce (trainner_amd/dp.py), so `gpu_ids` only selects the device;
  * kinds outside the SR hot path raise NotImplementedError instead of importing other archs.
"""
import functools
from collections import OrderedDict
import logging

import torch.nn as nn
from torch.nn import init

from ..options.options import opt_get  # noqa: F401  (re-exported like the reference)

logger = logging.getLogger("base")


def weights_init_kaiming(m, scale=1, bias_fill=0, **kwargs):
    """kaiming_normal_(a=0, fan_in) * scale on every Conv*/Linear* module, BatchNorm to (1, 0)
    (codes/models/networks.py:41-54)."""
    classname = m.__class__.__name__
    if hasattr(m, "weight") and (classname.find("Conv") != -1 or classname.find("Linear") != -1):
        init.kaiming_normal_(m.weight, **kwargs)
        m.weight.data *= scale
        if getattr(m, "bias", None) is not None:
            m.bias.data.fill_(bias_fill)
    elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
        init.constant_(m.weight, 1)
        if getattr(m, "bias", None) is not None:
            m.bias.data.fill_(bias_fill)


def init_weights(net, init_type="kaiming", scale=1, std=0.02, gain=0.02):
    logger.info("Initialization method [%s]", init_type)
    if init_type == "kaiming":
        net.apply(functools.partial(weights_init_kaiming, scale=scale))
    else:
        raise NotImplementedError("initialization method [{:s}] not implemented".format(init_type))


def get_network(opt, step=0, selector=None):
    opt_net = opt[selector]
    kind = opt_net.get("type").lower()
    opt_net_pass = dict(opt_net)
    init_type = opt_net_pass.pop("init_type", "kaiming")
    init_scale = opt_net_pass.pop("init_scale", 0.1)
    opt_net_pass.pop("strict", None)
    opt_net_pass.pop("type")

    if kind == "sr_resnet":
        from .modules.architectures import SRResNet_arch
        net = SRResNet_arch.SRResNet
    elif kind == "rrdb_net":
        from .modules.architectures import RRDBNet_arch
        net = RRDBNet_arch.RRDBNet
    elif kind == "discriminator_vgg":
        from .modules.architectures import discriminators
        net = discriminators.Discriminator_VGG
    elif kind == "unet":
        from .modules.architectures import discriminators
        net = discriminators.UNetDiscriminator
    elif kind == "resnet_net":
        from .modules.architectures import ResNet_arch
        net = ResNet_arch.ResnetGenerator
    elif kind == "unet_net":
        from .modules.architectures import UNet_arch
        net = UNet_arch.UnetGenerator
    elif kind in ("patchgan", "nlayerdiscriminator"):
        from .modules.architectures import discriminators
        net = discriminators.NLayerDiscriminator
    else:
        raise NotImplementedError("Model [{:s}] not recognized by the HIP engine".format(kind))

    net = net(**opt_net_pass)
    if opt["is_train"]:
        init_weights(net, init_type=init_type, scale=init_scale)
    return net


def define_network(opt, step=0, net_name="G"):
    return get_network(opt, step, "network_{}".format(net_name))


def define_G(opt, step=0, net_name="G"):
    return define_network(opt=opt, step=step, net_name=net_name)


def define_D(opt, net_name="D"):
    return define_network(opt=opt, net_name=net_name)


def define_F(opt):
    """Feature network for the perceptual loss (codes/models/networks.py:316-369)."""
    from .modules.architectures import perceptual
    z_norm = opt["datasets"]["train"].get("znorm", False)
    perc_opts = opt["train"].get("perceptual_opt")
    if perc_opts:
        net = perc_opts.get("feature_network", "vgg19")
        w_l_p = perc_opts.get("perceptual_layers", {"conv5_4": 1})
        w_l_s = perc_opts.get("style_layers", {})
        kw = dict(remove_pooling=perc_opts.get("remove_pooling", False),
                  use_input_norm=perc_opts.get("use_input_norm", True),
                  requires_grad=perc_opts.get("requires_grad", False),
                  change_padding=perc_opts.get("change_padding", False),
                  load_path=perc_opts.get("pretrained_path", None))
    else:
        net = opt["train"].get("feature_network", "vgg19") or "vgg19"
        w_l_p, w_l_s = {"conv5_4": 1}, {}
        kw = dict(remove_pooling=False, use_input_norm=True, requires_grad=False, change_padding=False, load_path=None)
    kw["allow_random_init"] = bool(opt["train"].get("perceptual_allow_random_init"))
    w_l = dict(w_l_p)
    w_l.update(w_l_s)
    if "resnet" in net:
        raise NotImplementedError("ResNet feature network is outside the SR hot path of the HIP engine")
    return perceptual.FeatureExtractor(listen_list=list(w_l.keys()), net=net, z_norm=bool(z_norm), pooling_stride=2, **kw)


# ESRGAN checkpoints exist in two key layouts: the "old arch" one this package (and the reference) uses
# (model.0 / model.1.sub.<i>.RDBk.convj.0 / model.3,6,8,10) and the "new arch" one of later community models
# (conv_first / RRDB_trunk.<i>.RDBk.convj / trunk_conv / upconv1,2 / HRconv / conv_last).  Same tensors, other
# names (codes/models/networks.py:400-481).  The trunk length is read from the keys, not assumed to be 23.
_NEW_HEAD = {"conv_first": "model.0", "upconv1": "model.3", "upconv2": "model.6", "HRconv": "model.8", "conv_last": "model.10"}


def mod2normal(state_dict):
    """new-arch -> old-arch ESRGAN keys (no-op for anything else)."""
    if "conv_first.weight" not in state_dict:
        return state_dict
    nb = 1 + max(int(k.split(".")[1]) for k in state_dict if k.startswith("RRDB_trunk."))
    out = OrderedDict()
    for k, v in state_dict.items():
        name, _, leaf = k.rpartition(".")                       # leaf: weight | bias
        if name in _NEW_HEAD:
            out["%s.%s" % (_NEW_HEAD[name], leaf)] = v
        elif name == "trunk_conv":
            out["model.1.sub.%d.%s" % (nb, leaf)] = v
        elif name.startswith("RRDB_trunk."):
            out["model.1.sub.%s.0.%s" % (name[len("RRDB_trunk."):], leaf)] = v
        else:
            raise KeyError("unexpected key %r in a new-arch ESRGAN checkpoint" % k)
    return out


def normal2mod(state_dict):
    """old-arch -> new-arch ESRGAN keys (for exporting to tools that expect the new layout)."""
    if "model.0.weight" not in state_dict:
        return state_dict
    old_head = {v: k for k, v in _NEW_HEAD.items()}
    trunk = [k for k in state_dict if k.startswith("model.1.sub.")]
    nb = max(int(k.split(".")[3]) for k in trunk)               # index of the trunk's closing conv
    out = OrderedDict()
    for k, v in state_dict.items():
        name, _, leaf = k.rpartition(".")
        if name in old_head:
            out["%s.%s" % (old_head[name], leaf)] = v
        elif name == "model.1.sub.%d" % nb:
            out["trunk_conv.%s" % leaf] = v
        elif name.startswith("model.1.sub.") and name.endswith(".0"):
            out["RRDB_trunk.%s.%s" % (name[len("model.1.sub."):-2], leaf)] = v
        else:
            raise KeyError("unexpected key %r in an old-arch ESRGAN checkpoint" % k)
    return out


def model_val(opt_net=None, state_dict=None, model_type=None):
    """Checkpoint key validation on load (codes/models/networks.py:483-497): a generator of type
    rrdb_net / esrgan accepts new-arch community checkpoints by renaming them to this package's layout."""
    if model_type == "G":
        kind = str(((opt_net or {}).get("network_G") or {}).get("type", "")).lower()
        if kind in ("rrdb_net", "esrgan"):
            return mod2normal(state_dict)
    return state_dict
