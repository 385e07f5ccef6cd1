"""YAML / JSON options loader of the SR training path.

Behavioural restatement of codes/options/options.py for the keys the SR path reads: `parse` (:539-631)
with NoneDict semantics (:52-69: a missing key reads as None), the scientific-notation aware YAML
reader (:83-110), `//`-commented JSON (:72-80), dataset path normalisation (:323-346, :508-536),
experiment paths and debug overrides (:571-597), network defaults (options/defaults.py), `*_rel`
schedule expansion (:612-624), CUDA_VISIBLE_DEVICES export (:626-629), `opt_get` (:647-659) and
`check_resume` (:670-714).  The augmentation-preset merging (:148-321, :366-506) configures the CPU
dataloader, which is outside this engine: those keys pass through untouched.
"""
import json
import logging
import os
import re
from collections import OrderedDict

from .defaults import get_network_defaults

logger = logging.getLogger("base")


class NoneDict(dict):
    """dict whose missing keys read as None."""

    def __missing__(self, key):
        return None


def dict_to_nonedict(opt):
    if isinstance(opt, dict):
        return NoneDict(**{k: dict_to_nonedict(v) for k, v in opt.items()})
    if isinstance(opt, list):
        return [dict_to_nonedict(v) for v in opt]
    return opt


def read_json(json_path):
    lines = []
    with open(json_path, "r") as f:
        for line in f:
            lines.append(line.split("//")[0])
    return json.loads("\n".join(lines), object_pairs_hook=OrderedDict)


_FLOAT_RE = re.compile(r"""^(?:
     [-+]?(?:[0-9][0-9_]*)\.[0-9_]*(?:[eE][-+]?[0-9]+)?
    |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
    |\.[0-9_]+(?:[eE][-+]?[0-9]+)?
    |[-+]?[0-9][0-9_]*(?::[0-5]?[0-9])+\.[0-9_]*
    |[-+]?\.(?:inf|Inf|INF)
    |\.(?:nan|NaN|NAN))$""", re.X)


def read_yaml(yaml_path):
    """Safe loader, ordered mappings, and `1e-4` / `5e5` resolved as floats like the reference."""
    import yaml

    class _Loader(getattr(yaml, "CSafeLoader", yaml.SafeLoader)):
        pass

    _Loader.add_constructor(yaml.resolver.BaseResolver.DEFAULT_MAPPING_TAG,
                            lambda loader, node: OrderedDict(loader.construct_pairs(node)))
    _Loader.add_implicit_resolver("tag:yaml.org,2002:float", _FLOAT_RE, list("-+0123456789."))
    with open(yaml_path, "r") as f:
        return yaml.load(f, Loader=_Loader)


_PATH_KEYS = ("HR", "HR_bg", "LR", "A", "B", "AB", "lq", "gt", "ref")


def parse_datasets(opt, scale=1):
    bm = opt.get("batch_multiplier", None)
    for phase_key, ds in opt["datasets"].items():
        phase = phase_key.split("_")[0]
        ds["phase"], ds["scale"] = phase, scale
        is_lmdb = False
        for key in _PATH_KEYS:
            p = ds.get("dataroot_" + key, None)
            if p is None:
                continue
            if isinstance(p, str):
                is_lmdb = os.path.splitext(p)[1].lower() == ".lmdb"
                p = [p]
            if not isinstance(p, list):
                raise ValueError("Unexpected path type: {}. Either a single path or a list of paths are "
                                 "supported.".format(type(p)))
            p = [os.path.normpath(os.path.expanduser(x)) for x in p]
            ds["dataroot_" + key] = p[0] if len(p) == 1 else p
        ds["data_type"] = "lmdb" if is_lmdb else "img"
        if ds.get("HR_size", None):
            ds["crop_size"] = ds["HR_size"]
        if phase == "train" and bm:
            ds["virtual_batch_size"] = bm * ds["batch_size"]
        if ds.get("virtual_batch_size", None):
            ds["virtual_batch_size"] = max(ds["virtual_batch_size"], ds["batch_size"])
        if phase == "train" and ds.get("subset_file") is not None:
            ds["subset_file"] = os.path.normpath(os.path.expanduser(ds["subset_file"]))
        if phase == "train" and scale != 1 and not ds.get("pre_crop", None) and not ds.get("preprocess"):
            ds["preprocess"] = "crop"
        if phase == "train" and not ds.get("dataroot_LR") and (ds.get("augs_strategy") or any(ds.get("add_%s_preset" % k) for k in ("blur", "resize", "noise"))):
            # presets overlay (options.py:148-165,366-463): the merged degradation configuration of the device pipeline
            # (dataops/degradations.degradation_config reads <presets_root>/<name>_{blur,resize,noise}.yaml and the dataset's overrides)
            from ..dataops.degradations import degradation_config
            # (only when the LR side is generated on the fly: with dataroot_LR the device pipeline is not used and nothing is merged)
            ds["presets_root"] = opt.get("presets_root", None) or ds.get("presets_root", None) or "presets"
            ds["_options_dir"] = opt.get("_options_dir")
            ds["degradation"] = degradation_config(ds, ds["presets_root"])
        ds.setdefault("resize_strat", "pre")
        if ds.get("tensor_shape", None):
            opt["tensor_shape"] = ds["tensor_shape"]
    return opt


def parse(opt_path, is_train=True):
    if not os.path.isfile(opt_path):
        alt = os.path.join("options", "train" if is_train else "test", opt_path)
        if not os.path.isfile(alt):
            raise ValueError("Configuration file {} not found.".format(alt))
        opt_path = alt
    ext = os.path.splitext(opt_path)[1].lower()
    if ext == ".json":
        opt = read_json(opt_path)
    elif ext in (".yml", ".yaml"):
        opt = read_yaml(opt_path)
    else:
        raise ValueError("Unknown configuration format: {}".format(ext))

    opt["is_train"] = is_train
    scale = opt.get("scale", 1)
    opt["_options_dir"] = os.path.dirname(os.path.abspath(opt_path))      # presets are also looked up next to the options file
    opt = parse_datasets(opt, scale)

    for key, path in opt["path"].items():
        if path:
            opt["path"][key] = os.path.normpath(os.path.expanduser(path))

    if is_train:
        root = os.path.join(opt["path"]["root"], "experiments", opt["name"])
        opt["path"].update(experiments_root=root, models=os.path.join(root, "models"),
                           training_state=os.path.join(root, "training_state"), log=root,
                           val_images=os.path.join(root, "val_images"))
        tr, lg = opt["train"], opt.setdefault("logger", OrderedDict())
        if tr.get("display_freq", None):
            opt["path"]["disp_images"] = os.path.join(root, "disp_images")
        tr["overwrite_val_imgs"] = tr.get("overwrite_val_imgs", None)
        tr["val_comparison"] = tr.get("val_comparison", None)
        lg["overwrite_chkp"] = lg.get("overwrite_chkp", None)
        if tr.get("use_frequency_separation", None) and not tr.get("fs", None):
            tr["fs"] = tr["use_frequency_separation"]
        if "debug" in opt["name"]:          # the authors' smoke-test mode
            tr["val_freq"], lg["print_freq"], tr["lr_decay_iter"] = 8, 2, 10
            lg["save_checkpoint_freq"] = 10000000 if "debug_nochkp" in opt["name"] else 8
    else:
        root = os.path.join(opt["path"]["root"], "results", opt["name"])
        opt["path"].update(results_root=root, log=root)

    opt = get_network_defaults(opt, is_train)

    if "train" in opt:
        tr = opt["train"]
        niter = tr["niter"]
        for k in ("T_period", "restarts", "lr_steps", "lr_steps_inverse"):
            if k + "_rel" in tr:
                tr[k] = [int(x * niter) for x in tr.pop(k + "_rel")]
        for k in ("swa_start_iter", "atg_start_iter"):
            if k + "_rel" in tr:
                tr[k] = int(tr.pop(k + "_rel") * niter)

    gpu_list = ",".join(str(x) for x in (opt.get("gpu_ids") or []))
    # one process per GPU (torchrun sets LOCAL_RANK): the launcher owns device visibility then
    if "LOCAL_RANK" not in os.environ:
        os.environ["CUDA_VISIBLE_DEVICES"] = gpu_list
        print("export CUDA_VISIBLE_DEVICES=" + gpu_list)
    return dict_to_nonedict(opt)


def dict2str(opt, indent_l=1):
    msg = ""
    for k, v in opt.items():
        pad = " " * (indent_l * 2)
        if isinstance(v, dict):
            msg += pad + k + ":[\n" + dict2str(v, indent_l + 1) + pad + "]\n"
        else:
            msg += pad + k + ": " + str(v) + "\n"
    return msg


def opt_get(opt=None, keys=None, default=None):
    if opt is None:
        return default
    ret = opt
    for k in keys or []:
        ret = ret.get(k, None)
        if ret is None:
            return default
    return ret


def check_resume(opt, resume_iter=None):
    """Point the pretrain_model_* paths at the <iter>_<net>.pth files next to a .state file (options.py:661-714): _G / _D, for
    CycleGAN _G_A, _G_B / _D_A, _D_B; the discriminators only when `train.gan_weight` is set."""
    state = opt["path"]["resume_state"]
    if not state:
        return
    opt["path"]["resume_state"] = os.path.normpath(state)
    keys = ["_A", "_B"] if opt.get("model") == "cyclegan" else [""]
    for ptype in ("_G", "_D"):
        for sufx in set([""] + keys):
            if opt["path"].get("pretrain_model" + ptype + sufx):
                logger.warning("pretrain_model%s%s path ignored, resuming training from a .state file.", ptype, sufx)
    idx = resume_iter if resume_iter else os.path.basename(opt["path"]["resume_state"]).split(".")[0]
    targets = ["_G"] + (["_D"] if opt["train"]["gan_weight"] else [])
    for ptype in targets:
        for mkey in keys:
            path = os.path.normpath(os.path.join(opt["path"]["models"], "{}{}{}.pth".format(idx, ptype, mkey)))
            opt["path"]["pretrain_model" + ptype + mkey] = path
            logger.info("Set [pretrain_model%s%s] to %s", ptype, mkey, path)
