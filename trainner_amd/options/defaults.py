"""Pre-flight expansion of short network names into full constructor dicts.

Restates the SR-path branches of codes/options/defaults.py: `esrgan`/`rrdb_net` (:36-63, incl. the
-lite/-mid presets), `sr_resnet`/`srresnet`/`srgan` (:98-112) and the `discriminator_vgg*` family
(:342-360; D `size` defaults to the training crop size).  Produces the same keys and defaults so a
train_sr.yml written for the reference parses to the same network dicts.
"""

_RRDB_KINDS = {"rrdb_net": (64, 23), "esrgan": (64, 23), "esrgan-lite": (32, 12), "esrgan-anime-lite": (64, 6),
               "esrgan-mid": (64, 6)}
_SRRESNET_KINDS = ("sr_resnet", "srresnet", "srgan")


def _kind_and_dict(net, which):
    if isinstance(net, str):
        return net.lower(), {}
    if isinstance(net, dict):
        net = dict(net)
        key = which if which in net else "type"
        return str(net[key]).lower(), net
    raise ValueError("network option must be a name or a dict")


def get_network_G_config(network_G, scale, crop_size):
    kind, src = _kind_and_dict(network_G, "which_model_G")
    full = {"strict": src.pop("strict", False)}
    if kind in _RRDB_KINDS:
        nf, nb = _RRDB_KINDS[kind]
        full["type"] = "rrdb_net"
        full["norm_type"] = src.pop("norm_type", None)
        full["mode"] = src.pop("mode", "CNA")
        full["nf"] = src.pop("nf", nf)
        full["nb"] = src.pop("nb", nb)
        full["nr"] = src.pop("nr", 3)
        full["in_nc"] = src.pop("in_nc", 3)
        full["out_nc"] = src.pop("out_nc", 3)
        full["gc"] = src.pop("gc", 32)
        full["convtype"] = src.pop("convtype", "Conv2D")
        full["act_type"] = src.pop("net_act", None) or src.pop("act_type", "leakyrelu")
        full["gaussian_noise"] = src.pop("gaussian", True)
        full["plus"] = src.pop("plus", False)
        full["finalact"] = src.pop("finalact", None)
        full["upscale"] = src.pop("scale", scale)
        full["upsample_mode"] = src.pop("upsample_mode", "upconv")
    elif kind in _SRRESNET_KINDS:
        full["type"] = "sr_resnet"
        full["in_nc"] = src.pop("in_nc", 3)
        full["out_nc"] = src.pop("out_nc", 3)
        full["nf"] = src.pop("nf", 64)
        full["nb"] = src.pop("nb", 16)
        full["upscale"] = src.pop("scale", scale)
        full["norm_type"] = src.pop("norm_type", None)
        full["act_type"] = src.pop("net_act", None) or src.pop("act_type", "relu")
        full["mode"] = src.pop("mode", "CNA")
        full["upsample_mode"] = src.pop("upsample_mode", "pixelshuffle")
        full["convtype"] = src.pop("convtype", "Conv2D")
        full["finalact"] = src.pop("finalact", None)
        full["res_scale"] = src.pop("res_scale", 1)
    elif "unet" in kind and "wbc" not in kind:            # pix2pix U-Net generator (defaults.py:194-217)
        full["type"] = "unet_net"
        full["input_nc"] = src.pop("in_nc", 3)
        full["output_nc"] = src.pop("out_nc", 3)
        full["num_downs"] = src.pop("num_downs", 7 if kind == "unet_128" else 8)
        want = {7: 128, 8: 256, 9: 512}.get(full["num_downs"])
        if want is not None:
            assert crop_size == want, f"Invalid crop size {crop_size} for UNET config, must be {want}"
        full["ngf"] = src.pop("ngf", 64)
        full["norm_type"] = src.pop("norm_type", "batch")
        full["use_dropout"] = src.pop("use_dropout", False)
        full["upsample_mode"] = src.pop("upsample_mode", "deconv")
    elif "resnet" in kind and kind != "sr_resnet":        # image-to-image ResNet generator (defaults.py:218-234)
        full["type"] = "resnet_net"
        full["input_nc"] = src.pop("in_nc", 3)
        full["output_nc"] = src.pop("out_nc", 3)
        full["n_blocks"] = src.pop("n_blocks", 6 if kind == "resnet_6blocks" else 9)
        full["ngf"] = src.pop("ngf", 64)
        full["norm_type"] = src.pop("norm_type", "instance")
        full["use_dropout"] = src.pop("use_dropout", False)
        full["upsample_mode"] = src.pop("upsample_mode", "deconv")
        full["padding_type"] = src.pop("padding_type", "reflect")
    else:
        raise NotImplementedError("Generator model [{}] is outside the SR hot path of the HIP engine".format(kind))
    for k in ("type", "which_model_G"):
        src.pop(k, None)
    if src:
        print(src)   # unprocessed keys, as the reference reports them
    return full


def get_network_D_config(network_D, scale, crop_size, model_G):
    arch = "PPON" if model_G == "ppon" else "ESRGAN"
    kind, src = _kind_and_dict(network_D, "which_model_D")
    full = {"strict": src.pop("strict", True)}
    if kind == "discriminator_vgg":
        src.pop("which_model_D", None)
        full["type"] = src.pop("type", kind)
        full["in_nc"] = src.pop("in_nc", 3)
        full["base_nf"] = src.pop("nf", 64)
        full["norm_type"] = src.pop("norm_type", "batch")
        full["mode"] = src.pop("mode", "CNA")
        full["act_type"] = src.pop("net_act", None) or src.pop("act_type", "leakyrelu")
        full["convtype"] = src.pop("convtype", "Conv2D")
        full["arch"] = src.pop("G_arch", arch)
        full["size"] = src.pop("D_size", crop_size)
    elif kind in ("patchgan", "nlayerdiscriminator"):  # PatchGAN (defaults.py:361-376)
        src.pop("which_model_D", None)
        src.pop("type", None)
        full["type"] = "patchgan"
        full["input_nc"] = src.pop("in_nc", 3)
        full["ndf"] = src.pop("nf", 64)
        full["n_layers"] = src.pop("n_layers", None) or src.pop("nlayer", 3)
        full["get_feats"] = src.pop("get_feats", False)
        full["patch"] = src.pop("patch_output", True)
        full["use_spectral_norm"] = src.pop("spectral_norm", None) or src.pop("use_spectral_norm", False)
    elif "unet" in kind:                               # Real-ESRGAN's U-Net discriminator (defaults.py:378-382)
        src.pop("which_model_D", None)
        src.pop("type", None)
        full["type"] = "unet"
        full["input_nc"] = src.pop("in_nc", 3)
        full["nf"] = src.pop("nf", 64)
        full["skip_connection"] = src.pop("skip_connection", True)
    else:
        raise NotImplementedError("Discriminator model [{}] is outside the SR hot path of the HIP engine".format(kind))
    if src:
        print(src)
    return full


def get_network_defaults(opt, is_train):
    scale = opt.get("scale", 1)
    crop_size = int(opt["datasets"]["train"]["crop_size"]) if is_train else opt.get("img_size")
    network_G = get_network_G_config(opt.pop("network_G", None), scale, crop_size)
    opt["network_G"] = network_G
    if opt.get("network_D", None):
        opt["network_D"] = get_network_D_config(opt.pop("network_D"), scale, crop_size, network_G["type"])
    return opt
