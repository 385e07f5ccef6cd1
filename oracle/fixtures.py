"""Helpers shared by the parity tests: rebuild a fixture's initial state and compare probes.

TEST INFRASTRUCTURE ONLY (see oracle/sr_oracle.py header).
"""
import os
import re
from collections import OrderedDict

import torch

from . import detrand
from . import sr_oracle

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def load(name):
    return torch.load(os.path.join(GOLDEN_DIR, name + ".pt"), map_location="cpu", weights_only=False)


def initial_state(keys, seed, **fill_kw):
    sd = OrderedDict()
    for k, shape in keys:
        dt = torch.int64 if k.endswith("num_batches_tracked") else torch.float32
        sd[k] = torch.zeros(shape, dtype=dt)
    return detrand.fill_state_dict_(sd, seed, **fill_kw)


def vgg_state(seed):
    sd = sr_oracle.vgg19_seeded_state()            # only for keys/shapes
    return detrand.fill_state_dict_(sd, seed, gain=1.0, bias_amp=0.05)


def initial_states(fx):
    g = initial_state(fx["g_keys"], fx["seeds"]["G"], **({"gain": fx["spec"]["g_gain"]} if "g_gain" in fx["spec"] else {}))
    d = initial_state(fx["d_keys"], fx["seeds"]["D"]) if fx["d_keys"] else None
    f = vgg_state(fx["seeds"]["F"]) if fx["spec"]["yaml"].get("feature", True) else None
    return g, d, f


def batches(fx):
    y = fx["spec"]["yaml"]
    for s in range(1, fx["spec"]["steps"] + 1):
        yield s, detrand.synthetic_pair(y["batch"], y["crop"], fx["seeds"]["data"] + s)


def oracle_for(fx):
    y = fx["spec"]["yaml"]
    g, d, f = initial_states(fx)
    ng = fx["network_G"]
    nd = fx["network_D"]
    unet = bool(nd) and nd["type"] == "unet"
    return sr_oracle.OracleSRStep(
        g, d, f, arch=ng["type"], nb=ng["nb"], d_size=(nd["size"] if nd and not unet else 0),
        d_nf=((nd["nf"] if unet else nd["base_nf"]) if nd else 0), d_arch=("unet" if unet else "discriminator_vgg"),
        pixel_weight=y.get("pixel_weight", 1e-2),
        feature_weight=1.0 if y.get("feature", True) else 0.0,
        gan_weight=5e-3 if y.get("gan", True) else 0.0,
        upsample_mode=ng.get("upsample_mode", "upconv"))


def probe_error(t, pr):
    """max |sample diff| / (max |sample| + tiny) and relative L2-norm error against a probe."""
    f = t.detach().flatten().to(torch.float64).cpu()
    s = f[::pr["stride"]][:pr["samples"].numel()]
    scale = pr["samples"].abs().max().item() + 1e-12
    e_s = (s - pr["samples"]).abs().max().item() / scale
    e_n = abs(f.norm().item() - pr["l2"]) / (pr["l2"] + 1e-12)
    return e_s, e_n


def bn_shadowed_biases(keys):
    """Conv biases that feed a BatchNorm (Discriminator_VGG, discriminators.py:24-34): their true
    gradient is exactly zero (BN subtracts the batch mean), so what any implementation computes
    is rounding noise, which Adam turns into +-lr updates of arbitrary sign.  They are excluded
    from post-step weight comparisons (their forward effect is nil as well)."""
    names = [k for k, _ in keys]
    out = set()
    for k in names:
        m = re.match(r"features\.(\d+)\.bias$", k)
        if m and ("features.%d.running_mean" % (int(m.group(1)) + 1)) in names:
            out.add(k)
    return out


def state_error(sd, probes, skip=(), lr_steps=1e-4):
    """Post-step weights vs probes, in units of the largest possible Adam displacement
    (lr * steps).  Adam divides each gradient element by its own running magnitude, so elements
    whose gradient is rounding noise move by +-lr whatever the implementation; we therefore
    report (worst |dp|/(lr*steps), mean |dp|/(lr*steps), worst key) and the tests bound the
    mean tightly and the worst loosely."""
    worst, worst_k, tot, cnt = 0.0, None, 0.0, 0
    for k, pr in probes.items():
        if k in skip or k.endswith("num_batches_tracked") or ".running_" in k:
            continue      # buffers are not Adam-driven: see buffers_error
        f = sd[k].detach().flatten().to(torch.float64).cpu()
        s_ = f[::pr["stride"]][:pr["samples"].numel()]
        d = (s_ - pr["samples"]).abs() / lr_steps
        tot += d.sum().item()
        cnt += d.numel()
        if d.max().item() > worst:
            worst, worst_k = d.max().item(), k
    return worst, tot / max(cnt, 1), worst_k


def buffers_error(sd, probes):
    """BatchNorm running statistics: plain relative comparison (not Adam-driven)."""
    worst = (0.0, None)
    for k, pr in probes.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            e_s, e_n = probe_error(sd[k].float(), pr)
            if max(e_s, e_n) > worst[0]:
                worst = (max(e_s, e_n), k)
    return worst


# --------------------------------------------------------------------------------------------------------------
# image-to-image fixtures (oracle/make_golden_i2i.py)
# --------------------------------------------------------------------------------------------------------------
def i2i_initial_states(fx):
    return {n: initial_state(fx["keys"][n], fx["seeds"][n]) for n in fx["model_names"]}


def i2i_batches(fx):
    y = fx["spec"]["yaml"]
    for s in range(1, fx["spec"]["steps"] + 1):
        sd = fx["seeds"]["data"] + s
        yield s, (detrand.uniform((y["batch"], 3, y["crop"], y["crop"]), sd, -1.0, 1.0),
                  detrand.uniform((y["batch"], 3, y["crop"], y["crop"]), sd + 5000, -1.0, 1.0))


def i2i_oracle_for(fx):
    from . import i2i_oracle
    y, st = fx["spec"]["yaml"], i2i_initial_states(fx)
    common = dict(n_blocks=fx["network_G"].get("n_blocks"), norm=fx["network_G"]["norm_type"], gan_type=y.get("gan_type", "vanilla"),
                  pixel_weight=y["pixel_weight"])
    if y["model"] == "pix2pix":
        orc = i2i_oracle.OraclePix2PixStep(st["G"], st["D"], **common)
    else:
        orc = i2i_oracle.OracleCycleGANStep(st["G_A"], st["G_B"], st["D_A"], st["D_B"], lambda_identity=y.get("lambda_identity"),
                                            pool_size=y.get("pool_size", 0), **common)
    if fx["network_G"]["type"] == "unet_net":
        orc.arch, orc.num_downs = "unet_net", fx["network_G"]["num_downs"]
    orc.form = y.get("gan_form", "standard") or "relativistic"      # (None = no gan_opt in the recipe = the reference's default)
    return orc


def norm_shadowed_biases(keys, norm_type):
    """ResnetGenerator conv biases in front of an InstanceNorm (every conv bias but the last layer's, ResNet_arch.py:47-90):
    exactly-zero true gradient, so Adam moves them by +-lr of arbitrary sign in any implementation (cf. bn_shadowed_biases)."""
    if norm_type != "instance":
        return set()
    biases = [k for k, _ in keys if k.endswith(".bias")]
    return set(biases[:-1])


def shipped_recipe_tree(rel, root):
    """tests/golden/shipped_recipes.json[rel] (oracle/make_golden_options.py: the option tree of the reference's
    codes/options/<rel>, locations as @ROOT@/...) with the placeholder replaced by `root`."""
    import json
    from collections import OrderedDict
    with open(os.path.join(GOLDEN_DIR, "shipped_recipes.json")) as f:
        tree = json.load(f, object_pairs_hook=OrderedDict)[rel]

    def sub(node):
        if isinstance(node, dict):
            return OrderedDict((k, sub(v)) for k, v in node.items())
        if isinstance(node, list):
            return [sub(v) for v in node]
        if isinstance(node, str) and node.startswith("@ROOT@/"):
            return os.path.join(root, node[len("@ROOT@/"):]) if node != "@ROOT@/" else root + "/"
        return node
    return sub(tree)


def write_recipe(rel, root, edit=None):
    """The recipe as a file `options.parse` reads: <root>/<basename of rel>, YAML or JSON by its extension.  `edit(tree)` may
    change the tree first (returns None or the new tree)."""
    import json
    tree = shipped_recipe_tree(rel, root)
    if edit is not None:
        tree = edit(tree) or tree
    path = os.path.join(root, os.path.basename(rel))
    os.makedirs(root, exist_ok=True)
    with open(path, "w") as f:
        if rel.endswith(".json"):
            json.dump(tree, f, indent=2)
        else:
            import yaml

            class Dumper(yaml.SafeDumper):
                pass

            Dumper.add_representer(type(tree), lambda d, data: d.represent_dict(data.items()))
            yaml.dump(tree, f, Dumper=Dumper, sort_keys=False, default_flow_style=False)
    return path
