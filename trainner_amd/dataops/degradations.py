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
RESRGAN = {
    # options/presets/resrgan_blur.yaml
    "blur": dict(types={"sinc": 0.1, "iso": 0.58, "aniso": 0.32}, prob=1.0, ks=(7, 21), sigma=(0.2, 3.0), angle=(-180, 180)),
    "blur2": dict(types={"sinc": 0.1, "iso": 0.58, "aniso": 0.32}, prob=0.8, ks=(7, 21), sigma=(0.2, 1.5), angle=(-180, 180)),
    "final_sinc_prob": 0.8,
    # options/presets/resrgan_resize.yaml
    "resize": dict(prob={"up": 0.2, "down": 0.7, "keep": 0.1}, up=(1.0, 1.5), down=(0.15, 1.0), algos=("area", "linear", "cubic")),
    "resize2": dict(prob={"up": 0.3, "down": 0.4, "keep": 0.3}, up=(1.0, 1.2), down=(0.3, 1.0), algos=("area", "linear", "cubic")),
    "final_algos": ("area", "linear", "cubic"),
    # options/presets/resrgan_noise.yaml
    "noise": dict(types=("gaussian", "poisson"), var=(1, 30), prob_color=0.6, multi=True, poisson_scale=(0.05, 3.0)),
    "noise2": dict(types=("gaussian", "poisson"), var=(1, 25), prob_color=0.6, multi=True, poisson_scale=(0.05, 2.5)),
    "jpeg": (30, 95),
}


class RealESRGANDegradation:
    """HR batch [N,3,H,W] (device, fp32 in [0,1], RGB) -> LR batch [N,3,H/scale,W/scale]."""

    def __init__(self, scale=4, preset=None, seed=0):
        self.scale = int(scale)
        self.p = dict(RESRGAN if preset is None else preset)
        self.rs = np.random.RandomState(seed)
        self.calls = 0

    # ---- parameter draws (per sample, like the reference's per-sample pipeline)
    def _blur_kernel(self, conf):
        rs = self.rs
        kind = rs.choice(list(conf["types"]), p=np.array(list(conf["types"].values())) / sum(conf["types"].values()))
        lo, hi = conf["ks"]
        if kind == "sinc":                       # RandomSincBlur.get_params: randint(lo, hi), forced odd
            ks = int(rs.randint(lo, hi))
            ks += 1 - ks % 2
            cutoff = rs.uniform(math.pi / 3 if ks < 13 else math.pi / 5, math.pi)
            return sinc_kernel(cutoff, ks)
        ks = int(rs.randint(lo, hi + 1))          # RandomAnIsoBlur.get_params: randint(lo, hi + 1), forced odd
        ks += 1 - ks % 2
        ks = min(ks, KMAX)
        sx = rs.uniform(*conf["sigma"])
        if kind == "iso":
            return gaussian_kernel(ks, (sx, sx))
        return gaussian_kernel(ks, (sx, rs.uniform(*conf["sigma"])), angle=rs.uniform(*conf["angle"]))

    def _resize_factor(self, conf):
        rs = self.rs
        mode = rs.choice(list(conf["prob"]), p=np.array(list(conf["prob"].values())) / sum(conf["prob"].values()))
        if mode == "up":
            return rs.uniform(*conf["up"])
        if mode == "down":
            return rs.uniform(*conf["down"])
        return 1.0

    def _noise(self, x, conf, seed):
        rs = self.rs
        kind = conf["types"][rs.randint(len(conf["types"]))]
        grey = rs.rand() >= conf["prob_color"]
        if kind == "gaussian":
            if conf["multi"] and rs.rand() > 0.66 and not grey:      # MC-AWGN a third of the time (transforms.py:1579-1586)
                sig = [math.sqrt(rs.uniform(*conf["var"])) for _ in range(3)]
            else:
                sig = [math.sqrt(rs.uniform(*conf["var"]))] * 3
            add_gaussian_noise(x, [sig], [grey], seed)
        else:
            add_poisson_noise(x, [rs.uniform(*conf["poisson_scale"])], [grey], seed)
        return x

    def _round(self, x, bkey, rkey, nkey, seed):
        p, rs = self.p, self.rs
        if rs.rand() < p[bkey]["prob"]:
            x = filter2d(x, [self._blur_kernel(p[bkey])])
        f = self._resize_factor(p[rkey])
        if f != 1.0:
            H, W = x.shape[2:]
            x = resize(x, (max(int(round(H * f)), 1), max(int(round(W * f)), 1)), p[rkey]["algos"][rs.randint(len(p[rkey]["algos"]))])
        x = self._noise(x, p[nkey], seed)
        return x

    def __call__(self, hr):
        hip.require_device(hr)
        N, C, H, W = hr.shape
        if C != 3 or H % self.scale or W % self.scale:
            raise ValueError("degradation expects RGB batches with sizes divisible by the scale")
        p, rs = self.p, self.rs
        out = torch.empty((N, 3, H // self.scale, W // self.scale), dtype=torch.float32, device=hr.device)
        self.calls += 1
        for n in range(N):
            seed = (self.calls << 20) + n * 16
            x = hr[n:n + 1].contiguous().clone()
            x = self._round(x, "blur", "resize", "noise", seed + 1)
            jpeg(x, [int(rs.randint(p["jpeg"][0], p["jpeg"][1] + 1))])
            x = self._round(x, "blur2", "resize2", "noise2", seed + 2)
            algo = p["final_algos"][rs.randint(len(p["final_algos"]))]
            final_sinc = rs.rand() < p["final_sinc_prob"]
            q = int(rs.randint(p["jpeg"][0], p["jpeg"][1] + 1))

            def final_resize(t):
                t = resize(t, (H // self.scale, W // self.scale), algo)
                if final_sinc:
                    ks = int(rs.randint(7, 21))
                    ks += 1 - ks % 2
                    t = filter2d(t, [sinc_kernel(rs.uniform(math.pi / 3 if ks < 13 else math.pi / 5, math.pi), ks)])
                return t

            if rs.rand() < 0.5:                  # jpeg then resize(+sinc), or the other way round (augmentations.py:1779-1784)
                x = final_resize(jpeg(x, [q]))
            else:
                x = jpeg(final_resize(x).clamp_(0, 1), [q])
            out[n] = x[0].clamp_(0, 1)
        return out
