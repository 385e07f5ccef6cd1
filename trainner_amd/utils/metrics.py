"""Validation metrics of the SR path, computed on the device (reference: codes/utils/metrics.py).

Same surface as the reference for what train.py uses (train.py:331-372,392-399): `MetricsDict(metrics='psnr,ssim')`,
`calculate_metrics(img1, img2, crop_size=4, only_y=False)` on tensor2np-style uint8 HWC images, `get_averages()`;
plus `calculate_psnr` / `calculate_ssim` on single images.  The images stay on the GPU as uint8 NHWC tensors
(`trainner_amd.dataops.common.tensor2np`), PSNR is an exact integer sum of squares and SSIM an fp64 Gaussian-window
reduction in libtrainner_hip.so (csrc/metrics.hip); one (sum, count) pair per image comes back to the host.
LPIPS needs a pretrained network and is outside the engine (raises).
"""
import math

import torch

from .. import hip
from ..ops import WS


def _as_batch_u8(img):
    """uint8 image(s) -> contiguous cuda uint8 [N,H,W,C] (accepts HW, HWC, NHWC; numpy arrays are uploaded)."""
    if not torch.is_tensor(img):
        img = torch.from_numpy(img)
    if img.dtype != torch.uint8:
        raise TypeError("metrics expect uint8 images (tensor2np output), got %s" % img.dtype)
    if img.dim() == 2:
        img = img[None, :, :, None]
    elif img.dim() == 3:
        img = img[None]
    elif img.dim() != 4:
        raise ValueError("Wrong input image dimensions.")
    hip.require_device()
    return img.to("cuda", non_blocking=True).contiguous()


def psnr_ssim_sums(img1, img2, crop=0, want_ssim=True):
    """-> float64 cpu tensor [N, 4]: sum of squared differences, count, SSIM-map sum, SSIM-map count (per image)."""
    a, b = _as_batch_u8(img1), _as_batch_u8(img2)
    if a.shape != b.shape:
        raise ValueError("Input images must have the same dimensions.")
    N, H, W, C = a.shape
    lib = hip.load()
    need = lib.tnr_metrics_workspace_bytes(N)
    ws = WS.get("metrics", need, a.device)
    out = torch.empty((N, 4), dtype=torch.float64, device=a.device)
    hip.check(lib.tnr_psnr_ssim_u8(a.data_ptr(), b.data_ptr(), N, H, W, C, int(crop), int(bool(want_ssim)), out.data_ptr(),
                                   ws.data_ptr(), ws.numel() * 8, hip.stream()), "psnr_ssim_u8")
    return out.cpu()


def _psnr_from(se, cnt):
    mse = se / cnt
    return float("inf") if mse == 0 else 20 * math.log10(255.0 / math.sqrt(mse))


def calculate_psnr(img1, img2, shave=4):
    """PSNR of one image pair in [0, 255] (utils/metrics.py:110-126)."""
    s = psnr_ssim_sums(img1, img2, crop=shave or 0, want_ssim=False)[0]
    return _psnr_from(float(s[0]), float(s[1]))


def calculate_ssim(img1, img2, shave=4):
    """SSIM of one image pair in [0, 255] (utils/metrics.py:180-223; the shave applies to 3-D inputs only)."""
    two_d = (img1.dim() if torch.is_tensor(img1) else img1.ndim) == 2
    s = psnr_ssim_sums(img1, img2, crop=0 if two_d else (shave or 0), want_ssim=True)[0]
    if float(s[3]) == 0:
        raise ValueError("image smaller than the 11 x 11 SSIM window")
    return float(s[2]) / float(s[3])


class MetricsDict:
    """utils/metrics.py:13-106 for 'psnr' and 'ssim'."""

    def __init__(self, metrics="psnr", lpips_model=None):
        names = [m.strip().lower() for m in metrics.split(",") if m.strip()]
        self.psnr, self.ssim = "psnr" in names, "ssim" in names
        if "lpips" in names:
            raise NotImplementedError("LPIPS validation metric is outside the SR hot path of the HIP engine")
        self.metrics_list = [{"name": n} for n in names]
        self.reset()

    def reset(self):
        self.count = 0
        self.psnr_sum = 0
        self.ssim_sum = 0

    def calculate_metrics(self, img1, img2, crop_size=4, only_y=False):
        if only_y:
            raise NotImplementedError("only_y metrics are not implemented by the HIP engine")
        s = psnr_ssim_sums(img1, img2, crop=crop_size, want_ssim=self.ssim)
        calculations = {}
        # one entry per image pair, like the reference (a batch counts as that many calls)
        for row in s:
            if self.psnr:
                calculations["psnr"] = _psnr_from(float(row[0]), float(row[1]))
                self.psnr_sum += calculations["psnr"]
            if self.ssim:
                calculations["ssim"] = float(row[2]) / float(row[3])
                self.ssim_sum += calculations["ssim"]
            self.count += 1
        return calculations

    def get_averages(self):
        out = {}
        if self.psnr:
            out["psnr"] = self.psnr_sum / self.count
        if self.ssim:
            out["ssim"] = self.ssim_sum / self.count
        self.reset()
        return out
