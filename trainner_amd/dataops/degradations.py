"""Real-ESRGAN degradation pipeline on the device.

The reference selects it with `augs_strategy: resrgan` on the train dataset (options/presets/README.md:25-33, presets
resrgan_{blur,resize,noise}.yaml) and runs it per sample on DataLoader worker CPUs through OpenCV
(dataops/augmentations.py:1666-1801: blur -> resize -> noise -> jpeg, a second round, final resize (+ sinc) and jpeg in
random order).  Here the HR batch is already on the GPU (data/feeder.py) and the LR batch is synthesised from it by
kernels (csrc/degrade.hip); only the random parameters and the <= 21 x 21 blur kernels are produced on the host:

    kernels        get_gaussian_kernel (augmennt/extra_functional.py:460-515), get_sinc_kernel (augmennt/spadd.py:16-37)
    parameters     RandomAnIsoBlur.get_params (augmennt/transforms.py:2546-2571), RandomSincBlur.get_params (:2619-2640),
                   RandomGaussianNoise.get_params (:1569-1600), resize probabilities / ranges of resrgan_resize.yaml

Differences from the reference, all deliberate: the images stay fp32 in [0, 1] between the stages (the reference rounds
to uint8 after every OpenCV call); noise comes from a counter-based generator seeded per batch (reproducible);
every sample of a batch draws its own parameters like the reference's per-sample pipeline, so intermediate sizes differ
per sample and the stages run per sample until the final resize makes the batch rectangular again.
"""
import math

import numpy as np
import torch

from .. import hip
from ..ops import WS

KMAX = 21


# ----------------------------------------------------------------------------------------------
# blur kernels (host, numpy)
# ----------------------------------------------------------------------------------------------
def norm_kernel(k):
    s = k.sum()
    if s == 0.0:
        k = k.copy()
        k[k.shape[0] // 2, k.shape[1] // 2] = 1
        return k
    return k / s


def rotate_kernel(kernel, angle):
    """cv2.getRotationMatrix2D((w // 2, h // 2), angle, 1) + cv2.warpAffine(kernel, M, (w, h)) with the defaults
    (INTER_LINEAR, BORDER_CONSTANT 0): output(x, y) samples the input at M^-1 (x, y), bilinear, zeros outside."""
    h, w = kernel.shape
    cx, cy = w // 2, h // 2
    a = math.radians(angle)
    al, be = math.cos(a), math.sin(a)
    # forward matrix M = [[al, be, (1-al) cx - be cy], [-be, al, be cx + (1-al) cy]]; warpAffine inverts it
    M = np.array([[al, be, (1 - al) * cx - be * cy], [-be, al, be * cx + (1 - al) * cy]], dtype=np.float64)
    A = np.linalg.inv(np.vstack([M, [0, 0, 1]]))
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    sx = A[0, 0] * xs + A[0, 1] * ys + A[0, 2]
    sy = A[1, 0] * xs + A[1, 1] * ys + A[1, 2]
    x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
    fx, fy = sx - x0, sy - y0

    def at(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        return np.where(ok, kernel[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)], 0.0)

    return ((1 - fy) * ((1 - fx) * at(y0, x0) + fx * at(y0, x0 + 1)) + fy * ((1 - fx) * at(y0 + 1, x0) + fx * at(y0 + 1, x0 + 1)))


def gaussian_kernel(kernel_size, sigma, angle=0.0, sf=1):
    """get_gaussian_kernel (extra_functional.py:460-515), 2-D: separable Gaussian on a float32 mesh grid, mean
    size // 2 - 0.5 (sf - 1), optional rotation, normalised to sum 1."""
    ks = [kernel_size] * 2
    sg = [sigma] * 2 if np.isscalar(sigma) else list(sigma)
    kernel = 1
    mesh = np.meshgrid(*(np.arange(s, dtype=np.float32) for s in ks))
    for size, std, m in zip(ks, sg, mesh):
        mean = size // 2 - 0.5 * (sf - 1)
        kernel = kernel * np.exp(-((m - mean) / std) ** 2 / 2.0)
        kernel = kernel / (std ** 2 * np.sqrt(2.0 * np.pi))
    if angle != 0:
        kernel = rotate_kernel(np.asarray(kernel, dtype=np.float64), angle)
    return norm_kernel(np.asarray(kernel, dtype=np.float64))


def sinc_kernel(cutoff, kernel_size, eps=1e-8):
    """get_sinc_kernel (spadd.py:16-37): circular low-pass cutoff * J1(cutoff r) / (2 pi r), centre cutoff^2 / (4 pi)."""
    from scipy.special import j1
    c = (kernel_size - 1) / 2
    x, y = np.mgrid[0:kernel_size, 0:kernel_size].astype(np.float64)
    r = np.sqrt((x - c) ** 2 + (y - c) ** 2)
    k = cutoff * j1(cutoff * r) / (2 * np.pi * r + eps)
    k[(kernel_size - 1) // 2, (kernel_size - 1) // 2] = cutoff ** 2 / (4 * np.pi)
    return norm_kernel(k)


def pad_kernel21(k):
    """centre a ks x ks kernel (ks odd <= 21) in the 21 x 21 slot tnr_filter2d reads."""
    ks = k.shape[0]
    if ks > KMAX or ks % 2 == 0 or k.shape[0] != k.shape[1]:
        raise ValueError("blur kernels are odd squares of at most %d taps" % KMAX)
    out = np.zeros((KMAX, KMAX), dtype=np.float32)
    o = (KMAX - ks) // 2
    out[o:o + ks, o:o + ks] = k
    return out


# ----------------------------------------------------------------------------------------------
# device ops (thin wrappers; fp32 NCHW in [0, 1])
# ----------------------------------------------------------------------------------------------
RESIZE_MODES = {"area": 0, "linear": 1, "bilinear": 1, "cubic": 2, "bicubic": 2}


def filter2d(img, kernels):
    """img [N,C,H,W]; kernels: one [ks,ks] array per image (or a [N,21,21] tensor)."""
    hip.require_device(img)
    img = img.contiguous()
    N, C, H, W = img.shape
    if not torch.is_tensor(kernels):
        kernels = torch.from_numpy(np.stack([pad_kernel21(np.asarray(k)) for k in kernels]))
    kd = kernels.to(img.device, torch.float32).contiguous()
    out = torch.empty_like(img)
    hip.check(hip.load().tnr_filter2d(img.data_ptr(), out.data_ptr(), kd.data_ptr(), N, C, H, W, hip.stream()), "filter2d")
    return out


def resize(img, size, mode="area"):
    hip.require_device(img)
    img = img.contiguous()
    N, C, H, W = img.shape
    Ho, Wo = int(size[0]), int(size[1])
    out = torch.empty((N, C, Ho, Wo), dtype=torch.float32, device=img.device)
    hip.check(hip.load().tnr_resize(img.data_ptr(), out.data_ptr(), N * C, H, W, Ho, Wo, RESIZE_MODES[mode], hip.stream()), "resize")
    return out


def _dev(a, dtype, device):
    return torch.as_tensor(np.asarray(a), dtype=dtype).to(device)


def add_gaussian_noise(img, sigma255, grey, seed, clip=True):
    """in place: sigma255 [N] or [N,C] (std in 0..255 units), grey [N] bool."""
    N, C, H, W = img.shape
    s = np.asarray(sigma255, dtype=np.float32)
    if s.ndim == 1:
        s = np.repeat(s[:, None], C, 1)
    ds, dg = _dev(s, torch.float32, img.device), _dev(np.asarray(grey, dtype=np.int32), torch.int32, img.device)   # (kept alive)
    hip.check(hip.load().tnr_noise_gaussian(img.data_ptr(), N, C, H, W, ds.data_ptr(), dg.data_ptr(), int(seed) & (2 ** 64 - 1),
                                           int(clip), hip.stream()), "noise_gaussian")
    return img


def poisson_levels(img):
    """vals = 2^ceil(log2(#distinct 8-bit levels)) per image (extra_functional.py:206-207), computed on the device."""
    q = (img.clamp(0, 1) * 255.0).round().to(torch.int64).flatten(1)
    out = []
    for row in q:
        n = int(torch.bincount(row, minlength=256).gt(0).sum().item())
        out.append(float(2 ** math.ceil(math.log2(max(n, 1)))) if n > 1 else 1.0)
    return out


def add_poisson_noise(img, scale, grey, seed, vals=None, clip=True):
    N, C, H, W = img.shape
    vals = poisson_levels(img) if vals is None else vals
    dv, dsc = _dev(vals, torch.float32, img.device), _dev(scale, torch.float32, img.device)
    dg = _dev(np.asarray(grey, dtype=np.int32), torch.int32, img.device)
    hip.check(hip.load().tnr_noise_poisson(img.data_ptr(), N, C, H, W, dv.data_ptr(), dsc.data_ptr(), dg.data_ptr(),
                                          int(seed) & (2 ** 64 - 1), int(clip), hip.stream()), "noise_poisson")
    return img


def jpeg(img, quality):
    """in place JPEG round trip of RGB images, quality [N] ints."""
    N, C, H, W = img.shape
    if C != 3:
        raise ValueError("jpeg simulation expects RGB images")
    lib = hip.load()
    need = lib.tnr_jpeg_workspace_bytes(N, H, W)
    ws = WS.get("jpeg@%x" % hip.stream(), need, img.device)
    dq = _dev(np.asarray(quality, dtype=np.int32), torch.int32, img.device)
    hip.check(lib.tnr_jpeg_sim(img.data_ptr(), N, H, W, dq.data_ptr(), ws.data_ptr(), ws.numel() * 8, hip.stream()), "jpeg_sim")
    return img


# ----------------------------------------------------------------------------------------------
# the pipeline
# ----------------------------------------------------------------------------------------------
# Built-in configuration = the values of the reference's resrgan presets (options/presets/resrgan_{blur,resize,noise}.yaml over
# base_*.yaml); `degradation_config` replaces any of it from the user's preset files / dataset options, like options.parse does
# in the reference (options/options.py:148-165,366-463; options/presets/README.md "Overriding").
RESRGAN = {
    "blur": dict(enabled=True, types={"sinc": 0.1, "iso": 0.58, "aniso": 0.32}, prob=1.0,
                 iso=dict(ks=(7, 21), sigmaX=(0.2, 3.0)), aniso=dict(ks=(7, 21), sigmaX=(0.2, 3.0), sigmaY=(0.2, 3.0), angle=(-180, 180)),
                 sinc=dict(ks=(7, 21))),
    "blur2": dict(enabled=True, types={"sinc": 0.1, "iso": 0.58, "aniso": 0.32}, prob=0.8,
                  iso=dict(ks=(7, 21), sigmaX=(0.2, 1.5)), aniso=dict(ks=(7, 21), sigmaX=(0.2, 1.5), sigmaY=(0.2, 1.5), angle=(-180, 180)),
                  sinc=dict(ks=(7, 21))),
    "final_blur": dict(enabled=True, prob=0.8, sinc=dict(ks=(7, 21))),
    "resize": dict(enabled=True, prob={"up": 0.2, "down": 0.7, "keep": 0.1}, up=(1.0, 1.5), down=(0.15, 1.0), algos=("area", "linear", "cubic")),
    "resize2": dict(enabled=True, prob={"up": 0.3, "down": 0.4, "keep": 0.3}, up=(1.0, 1.2), down=(0.3, 1.0), algos=("area", "linear", "cubic")),
    "final_scale": dict(enabled=True, algos=("area", "linear", "cubic")),
    "noise": dict(enabled=True, types=("gaussian", "poisson"), gaussian=dict(var=(1, 30), prob_color=0.6, multi=True),
                  poisson=dict(scale=(0.05, 3.0), prob_color=0.6)),
    "noise2": dict(enabled=True, types=("gaussian", "poisson"), gaussian=dict(var=(1, 25), prob_color=0.6, multi=True),
                   poisson=dict(scale=(0.05, 2.5), prob_color=0.6)),
    "compression": dict(enabled=True, quality=(30, 95)),
    "final_compression": dict(enabled=True, quality=(30, 95)),
    "shuffle": False,
}
# What the reference merges every preset OVER (options/presets/base_{blur,resize,noise}.yaml): every degradation stage is off -- only
# the plain down-scaling to the LR size (`lr_downscale: true`, linear / bicubic) is on -- and the per-type values are the base files'.
BASE = {
    "blur": dict(enabled=False, types={}, prob=1.0,
                 iso=dict(ks=(7, 21), sigmaX=(0.1, 2.8)), aniso=dict(ks=(7, 21), sigmaX=(0.1, 2.8), sigmaY=(0.1, 2.8), angle=(0, 180)),
                 sinc=dict(ks=(7, 21))),
    "blur2": dict(enabled=False, types={}, prob=1.0,
                  iso=dict(ks=(7, 21), sigmaX=(0.1, 2.8)), aniso=dict(ks=(7, 21), sigmaX=(0.1, 2.8), sigmaY=(0.1, 2.8), angle=(0, 180)),
                  sinc=dict(ks=(7, 21))),
    "final_blur": dict(enabled=False, prob=1.0, sinc=dict(ks=(7, 21))),
    # (base_resize.yaml's `lr_downscale: true` with no random range active = the plain HR -> LR down-scaling by `scale`: in this
    #  pipeline's terms the final scale; the random-factor resizes are what the resrgan resize preset switches on)
    "resize": dict(enabled=False, prob={"down": 1.0}, up=(1.0, 1.5), down=(0.15, 1.0), algos=("linear", "cubic")),
    "resize2": dict(enabled=False, prob={"down": 1.0}, up=(1.0, 1.2), down=(0.3, 1.0), algos=("linear", "cubic")),
    "final_scale": dict(enabled=True, algos=("linear", "cubic")),
    "noise": dict(enabled=False, types=(), gaussian=dict(var=(1, 25), prob_color=0.5, multi=True), poisson=dict(scale=(0.5, 1.0), prob_color=0.5)),
    "noise2": dict(enabled=False, types=(), gaussian=dict(var=(1, 25), prob_color=0.5, multi=True), poisson=dict(scale=(0.5, 1.0), prob_color=0.5)),
    "compression": dict(enabled=False, quality=(30, 95)),
    "final_compression": dict(enabled=False, quality=(30, 95)),
    "shuffle": False,
}
_KIND_KEYS = {"blur": ("blur", "blur2", "final_blur"), "resize": ("resize", "resize2", "final_scale"),
              "noise": ("noise", "noise2", "compression", "final_compression")}
_ALGO = {"area": "area", "linear": "linear", "bilinear": "linear", "cubic": "cubic", "bicubic": "cubic"}


def _algo(t):
    if str(t) not in _ALGO:
        raise NotImplementedError("resize type %r is not implemented by the device pipeline (area / linear / cubic)" % (t,))
    return _ALGO[str(t)]


def _load_preset(path, kind):
    import yaml
    with open(path) as f:
        doc = yaml.safe_load(f)
    if doc.get("kind") != kind:
        raise ValueError("preset %s: kind %r, expected %r" % (path, doc.get("kind"), kind))
    return doc.get("config") or {}


def find_preset_file(presets_root, name, options_dir=None):
    """options/options.py:166-181: `options/<presets_root>/<name>.yaml` relative to the working directory, or the bare file name;
    additionally `<presets_root>` itself (absolute or relative) and the same two places next to the options file."""
    if not name:
        return None
    import os
    full = name if name.endswith(".yaml") else name + ".yaml"
    root = presets_root or "presets"
    cands = [os.path.join("options", root, full), os.path.join(root, full), full]
    if options_dir:
        cands += [os.path.join(options_dir, root, full), os.path.join(options_dir, full), os.path.join(options_dir, "..", root, full)]
    for c in cands:
        if os.path.isfile(c):
            return c
    return None


def degradation_config(dataset_opt=None, presets_root=None):
    """The pipeline configuration of a train dataset with `augs_strategy: <name>` / `add_{blur,resize,noise}_preset` (reference:
    options/options.py presets_names :148-165, find_preset_file :166-181 and the merge :366-463).  Like there, everything starts
    from the base presets -- every degradation stage OFF, plain down-scaling on (BASE) -- and only what a named preset or a key in
    the dataset options switches on runs: `add_blur_preset: resrgan_blur` alone gives the Real-ESRGAN blurs and nothing else.
    Per kind: the preset file `<name>.yaml` (find_preset_file) when it exists; else, for the `resrgan_*` names, the built-in copy of
    the reference's resrgan_{blur,resize,noise}.yaml (logged); any other missing preset is skipped like in the reference
    (`realsr` ships no blur preset).  Then the keys given directly in the dataset options (pipeline switches such as
    `lr_blur_types`, `blur_prob`, `shuffle_degradations`, per-type dictionaries under `aug_configs`).  Types the device pipeline
    does not implement raise NotImplementedError.  No arguments: the full Real-ESRGAN configuration (`augs_strategy: resrgan`)."""
    import copy
    import logging
    log = logging.getLogger("base")
    if dataset_opt is None:
        dataset_opt = {"augs_strategy": "resrgan"}
    ds = dict(dataset_opt)
    cfg = copy.deepcopy(BASE)
    strat = ds.get("augs_strategy")
    names = {k: ds.get("add_%s_preset" % k) or (("%s_%s" % (strat, k)) if strat else None) for k in ("blur", "resize", "noise")}
    root = presets_root or ds.get("presets_root") or "presets"
    raw = {}
    for k, kind in (("blur", "Blur"), ("resize", "Resize"), ("noise", "Noise")):
        raw[k] = {}
        path = find_preset_file(root, names[k], ds.get("_options_dir"))
        if path:
            raw[k] = _load_preset(path, kind)
        elif names[k] and str(names[k]) == "resrgan_%s" % k:
            log.info("preset %s not found under %r: using the built-in copy of the reference's file", names[k], root)
            for key in _KIND_KEYS[k]:
                cfg[key] = copy.deepcopy(RESRGAN[key])
        elif names[k]:
            log.warning("preset %s not found under %r: skipped (as the reference does)", names[k], root)
    pipe = {}
    for k in ("blur", "resize", "noise"):
        pipe.update(raw[k].get("pipeline") or {})
    aug = {}
    for k in ("blur", "resize", "noise"):
        aug.update({n: v for n, v in raw[k].items() if n != "pipeline"})
    for n, v in (ds.get("aug_configs") or {}).items():           # per-type overrides from the options file
        aug[n] = dict(aug.get(n) or {}, **(v or {}))
    for n in ("lr_blur", "lr_blur_types", "blur_prob", "lr_blur2", "lr_blur_types2", "blur_prob2", "shuffle_degradations", "final_blur",
              "final_blur_prob", "lr_downscale", "lr_downscale_types", "lr_downscale2", "lr_downscale_types2", "final_scale",
              "final_scale_types", "lr_noise", "lr_noise_types", "lr_noise2", "lr_noise_types2", "compression", "final_compression"):
        if n in ds and ds[n] is not None:
            pipe[n] = ds[n]

    def weights(v):
        return {t: 1.0 for t in v} if isinstance(v, (list, tuple)) else dict(v)

    def rng(v, default):
        return default if v is None else (tuple(v) if isinstance(v, (list, tuple)) else (v, v))

    for key, sw, tk, pk, sfx in (("blur", "lr_blur", "lr_blur_types", "blur_prob", ""), ("blur2", "lr_blur2", "lr_blur_types2", "blur_prob2", "2")):
        c = cfg[key]
        if sw in pipe:
            c["enabled"] = bool(pipe[sw])
        if tk in pipe and pipe[tk]:
            c["types"] = weights(pipe[tk])
        if pk in pipe:
            c["prob"] = float(pipe[pk])
        for t in c["types"]:
            if t not in ("iso", "aniso", "sinc"):
                raise NotImplementedError("blur type %r is not implemented by the device pipeline (iso / aniso / sinc)" % t)
            a = aug.get(t + sfx) or {}
            d = c[t]
            if "kernel_size" in a or "min_kernel_size" in a:
                d["ks"] = (int(a.get("min_kernel_size", d["ks"][0])), int(a.get("kernel_size", d["ks"][1])))
            for src, dst in (("sigmaX", "sigmaX"), ("sigmaY", "sigmaY"), ("angle", "angle")):
                if dst in d and a.get(src) is not None:
                    d[dst] = rng(a[src], d[dst])
    if "final_blur" in pipe:
        cfg["final_blur"]["enabled"] = bool(pipe["final_blur"])
        if pipe["final_blur"] and list(pipe["final_blur"]) != ["sinc"]:
            raise NotImplementedError("final_blur %r: only [sinc] is implemented" % (pipe["final_blur"],))
    if "final_blur_prob" in pipe:
        cfg["final_blur"]["prob"] = float(pipe["final_blur_prob"])
    for key, sw, tk in (("resize", "lr_downscale", "lr_downscale_types"), ("resize2", "lr_downscale2", "lr_downscale_types2")):
        c = cfg[key]
        if sw in pipe:
            c["enabled"] = bool(pipe[sw])
        if tk in pipe and pipe[tk]:
            c["algos"] = tuple(_algo(t) for t in pipe[tk])
        a = aug.get(key) or {}
        if a.get("resize_prob"):
            c["prob"] = dict(a["resize_prob"])
        c["up"], c["down"] = rng(a.get("resize_range_up"), c["up"]), rng(a.get("resize_range_down"), c["down"])
    if "final_scale" in pipe:
        cfg["final_scale"]["enabled"] = bool(pipe["final_scale"])
    if pipe.get("final_scale_types"):
        cfg["final_scale"]["algos"] = tuple(_algo(t) for t in pipe["final_scale_types"])
    for key, sw, tk, sfx in (("noise", "lr_noise", "lr_noise_types", ""), ("noise2", "lr_noise2", "lr_noise_types2", "2")):
        c = cfg[key]
        if sw in pipe:
            c["enabled"] = bool(pipe[sw])
        if tk in pipe and pipe[tk]:
            c["types"] = tuple(pipe[tk])
        for t in c["types"]:
            if t not in ("gaussian", "poisson"):
                raise NotImplementedError("noise type %r is not implemented by the device pipeline (gaussian / poisson)" % t)
            a = aug.get(t + sfx) or {}
            if t == "gaussian":
                c[t]["var"] = rng(a.get("var_limit"), c[t]["var"])
                c[t]["multi"] = bool(a.get("multi", c[t]["multi"]))
            else:
                c[t]["scale"] = rng(a.get("scale_range"), c[t]["scale"])
            c[t]["prob_color"] = float(a.get("prob_color", c[t]["prob_color"]))
    for key in ("compression", "final_compression"):
        if key in pipe:
            v = pipe[key]
            cfg[key]["enabled"] = bool(v)
            if v and list(v) != ["jpeg"]:
                raise NotImplementedError("%s %r: only [jpeg] is implemented" % (key, v))
        a = aug.get("jpeg") or {}
        cfg[key]["quality"] = (int(a.get("min_quality", cfg[key]["quality"][0])), int(a.get("max_quality", cfg[key]["quality"][1])))
    cfg["shuffle"] = bool(pipe.get("shuffle_degradations", cfg["shuffle"]))
    return cfg


class RealESRGANDegradation:
    """HR batch [N,3,H,W] (device, fp32 in [0,1], RGB) -> LR batch [N,3,H/scale,W/scale].

    Two halves.  `plan(N, H, W)` draws, on the host and in the reference's order, every random decision of every sample and returns
    one operation list per sample -- ("blur", kernel), ("resize", (h, w), algo), ("gaussian", sigma[3], grey, seed),
    ("poisson", scale, grey, seed), ("jpeg", quality) -- following aug_pipeline (dataops/augmentations.py:1666-1801): blur -> resize ->
    noise -> compression -> blur2 -> resize2 -> noise2 (shuffled as a whole when `shuffle_degradations`), then final compression and
    final resize (+ sinc) in random order.  `run(hr, plans)` executes the lists on the device: operations at the same position that
    agree in kind and tensor shape across samples -- always the first blur over the full-size HR batch, the most expensive stage --
    run as ONE launch with per-sample parameters; after the random resizes the shapes differ and the rest runs per sample."""

    def __init__(self, scale=4, preset=None, seed=0):
        self.scale = int(scale)
        self.p = degradation_config() if preset is None else preset
        self.rs = np.random.RandomState(seed)
        self.calls = 0

    # ---- parameter draws (per sample, like the reference's per-sample pipeline)
    def _sinc_kernel(self, conf):
        rs = self.rs
        lo, hi = conf["ks"]
        ks = int(rs.randint(lo, hi))              # RandomSincBlur.get_params (transforms.py:2619-2640): randint(lo, hi), forced odd
        ks += 1 - ks % 2
        return sinc_kernel(rs.uniform(math.pi / 3 if ks < 13 else math.pi / 5, math.pi), ks)

    def _blur_kernel(self, conf):
        rs = self.rs
        kind = rs.choice(list(conf["types"]), p=np.array(list(conf["types"].values()), dtype=np.float64) / sum(conf["types"].values()))
        if kind == "sinc":
            return self._sinc_kernel(conf["sinc"])
        c = conf[kind]
        lo, hi = c["ks"]
        ks = int(rs.randint(lo, hi + 1))          # RandomAnIsoBlur.get_params (transforms.py:2546-2571): randint(lo, hi + 1), forced odd
        ks += 1 - ks % 2
        ks = min(ks, KMAX)
        sx = rs.uniform(*c["sigmaX"])
        if kind == "iso":
            return gaussian_kernel(ks, (sx, sx))
        return gaussian_kernel(ks, (sx, rs.uniform(*c["sigmaY"])), angle=rs.uniform(*c["angle"]))

    def _resize_op(self, conf, h, w):
        rs = self.rs
        mode = rs.choice(list(conf["prob"]), p=np.array(list(conf["prob"].values()), dtype=np.float64) / sum(conf["prob"].values()))
        f = rs.uniform(*conf["up"]) if mode == "up" else (rs.uniform(*conf["down"]) if mode == "down" else 1.0)
        if f == 1.0:
            return None
        return ("resize", (max(int(round(h * f)), 1), max(int(round(w * f)), 1)), conf["algos"][rs.randint(len(conf["algos"]))])

    def _noise_op(self, conf, seed):
        rs = self.rs
        kind = conf["types"][rs.randint(len(conf["types"]))]
        c = conf[kind]
        grey = bool(rs.rand() >= c["prob_color"])
        if kind == "gaussian":
            if c["multi"] and rs.rand() > 0.66 and not grey:          # MC-AWGN a third of the time (transforms.py:1579-1586)
                sig = [math.sqrt(rs.uniform(*c["var"])) for _ in range(3)]
            else:
                sig = [math.sqrt(rs.uniform(*c["var"]))] * 3
            return ("gaussian", sig, grey, seed)
        return ("poisson", float(rs.uniform(*c["scale"])), grey, seed)

    def _plan_one(self, H, W, seed):
        p, rs = self.p, self.rs
        ops, h, w = [], H, W

        def stage(bkey, rkey, nkey, sd):
            nonlocal h, w
            if p[bkey]["enabled"] and rs.rand() < p[bkey]["prob"]:
                ops.append(("blur", self._blur_kernel(p[bkey])))
            if p[rkey]["enabled"]:
                r = self._resize_op(p[rkey], h, w)
                if r is not None:
                    ops.append(r)
                    h, w = r[1]
            if p[nkey]["enabled"]:
                ops.append(self._noise_op(p[nkey], sd))

        stage("blur", "resize", "noise", seed + 1)
        if p["compression"]["enabled"]:
            ops.append(("jpeg", int(rs.randint(p["compression"]["quality"][0], p["compression"]["quality"][1] + 1))))
        stage("blur2", "resize2", "noise2", seed + 2)
        if p["shuffle"]:                                  # random.shuffle(transform_list) (augmentations.py:1749-1750)
            order = rs.permutation(len(ops))
            ops = [ops[i] for i in order]
            h, w = H, W
            fixed = []
            for op in ops:                                # a resize target was drawn as a FACTOR of its input: re-derive the sizes
                if op[0] == "resize":
                    raise NotImplementedError("shuffle_degradations with in-pipeline resizes is not implemented by the device pipeline")
                fixed.append(op)
            ops = fixed
        final_c = [("jpeg", int(rs.randint(p["final_compression"]["quality"][0], p["final_compression"]["quality"][1] + 1)))] \
            if p["final_compression"]["enabled"] else []
        final_r = []
        if p["final_scale"]["enabled"]:
            final_r.append(("resize", (H // self.scale, W // self.scale), p["final_scale"]["algos"][rs.randint(len(p["final_scale"]["algos"]))]))
            if p["final_blur"]["enabled"] and rs.rand() < p["final_blur"]["prob"]:
                final_r.append(("blur", self._sinc_kernel(p["final_blur"]["sinc"])))
        elif (h, w) != (H // self.scale, W // self.scale):
            raise ValueError("final_scale is off but the pipeline does not end at the LR size")
        if final_c and rs.rand() < 0.5:                   # compression then resize (+ sinc), or the other way round (augmentations.py:1779-1784)
            ops += final_c + final_r
        else:
            ops += final_r + final_c
        return ops

    def plan(self, N, H, W):
        self.calls += 1
        return [self._plan_one(H, W, (self.calls << 20) + n * 16) for n in range(N)]

    # ---- execution
    @staticmethod
    def _apply(x, ops):
        """x [n,3,h,w]; ops: one operation per image of x, all of the same kind (and target size)."""
        kind = ops[0][0]
        if kind == "blur":
            return filter2d(x, [o[1] for o in ops])
        if kind == "resize":
            algos = {o[2] for o in ops}
            if len(algos) == 1:
                return resize(x, ops[0][1], ops[0][2])
            return torch.cat([resize(x[i:i + 1], o[1], o[2]) for i, o in enumerate(ops)], 0)
        x = x.clamp_(0, 1) if kind == "jpeg" else x
        for i, o in enumerate(ops):        # noise / jpeg: in place, per-image parameters (seeds are per sample)
            xi = x[i:i + 1]
            if kind == "gaussian":
                add_gaussian_noise(xi, [o[1]], [o[2]], o[3])
            elif kind == "poisson":
                add_poisson_noise(xi, [o[1]], [o[2]], o[3])
            else:
                jpeg(xi, [o[1]])
        return x

    def run(self, hr, plans):
        N, C, H, W = hr.shape
        out = torch.empty((N, 3, H // self.scale, W // self.scale), dtype=torch.float32, device=hr.device)
        # leading operations shared by the whole batch (same kind, same shapes) run as one launch each
        x, pos = hr.contiguous().clone(), 0
        while all(len(pl) > pos for pl in plans) and len({pl[pos][0] for pl in plans}) == 1 and \
                (plans[0][pos][0] != "resize" or len({pl[pos][1] for pl in plans}) == 1):
            x = self._apply(x, [pl[pos] for pl in plans])
            pos += 1
        for n, pl in enumerate(plans):
            xn = x[n:n + 1].contiguous()
            for op in pl[pos:]:
                xn = self._apply(xn, [op])
            if tuple(xn.shape[2:]) != tuple(out.shape[2:]):
                raise hip.HipEngineError("degradation plan ended at %s, expected %s" % (tuple(xn.shape[2:]), tuple(out.shape[2:])))
            out[n] = xn[0].clamp_(0, 1)
        return out

    def __call__(self, hr):
        hip.require_device(hr)
        N, C, H, W = hr.shape
        if C != 3 or H % self.scale or W % self.scale:
            raise ValueError("degradation expects RGB batches with sizes divisible by the scale")
        return self.run(hr, self.plan(N, H, W))
