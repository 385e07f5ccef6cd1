"""CPU restatement of the reference's validation metrics (numpy; TEST INFRASTRUCTURE ONLY).

calculate_psnr: codes/utils/metrics.py:110-126 (importable from the reference: pure numpy) -- pinned by
tests/golden/metrics.pt, which oracle/make_golden_metrics.py produces with the REFERENCE's own tensor2np,
calculate_psnr and calculate_ssim.  ssim / calculate_ssim (codes/utils/metrics.py:180-223) call cv2.getGaussianKernel and
cv2.filter2D; cv2 is not installed here, so the fixture run serves those two calls from scipy (oracle/stubs/cv2) and executes the
reference's code around them; this module restates the same algorithm directly (11-tap Gaussian, sigma 1.5:
exp(-(i-5)^2 / (2 sigma^2)) normalised; 'valid' correlation; C1, C2 as in the reference) in float64 and must agree with the
fixture to 1e-12 (tests/test_metrics.py).
"""
import math

import numpy as np


def tensor2np(img, rgb2bgr=True, denormalize=False):
    """dataops/common.py:502-566 for a CHW float tensor / array in any range -> HWC uint8."""
    a = np.asarray(img, dtype=np.float32)
    if a.ndim == 3 and a.shape[0] in (3, 4) and rgb2bgr:
        a = a[[2, 1, 0] + ([3] if a.shape[0] == 4 else []), :, :]
    a = np.transpose(a, (1, 2, 0)) if a.ndim == 3 else a
    if denormalize:
        a = (a + 1.0) / 2.0
    return np.clip(255 * a, 0, 255).round().astype(np.uint8)


def calculate_psnr(img1, img2, shave=4):
    if shave:
        img1 = img1[shave:-shave, shave:-shave, ...]
        img2 = img2[shave:-shave, shave:-shave, ...]
    mse = np.mean((img1.astype(np.float64) - img2.astype(np.float64)) ** 2)
    return float("inf") if mse == 0 else 20 * math.log10(255.0 / math.sqrt(mse))


def gaussian_window():
    k = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2))
    k /= k.sum()
    return np.outer(k, k)


def _valid_filter(img, win):
    h, w = img.shape[0] - 10, img.shape[1] - 10
    out = np.zeros((h, w) + img.shape[2:], dtype=np.float64)
    for dy in range(11):
        for dx in range(11):
            out += win[dy, dx] * img[dy:dy + h, dx:dx + w, ...]
    return out


def ssim_map(img1, img2):
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    a, b = img1.astype(np.float64), img2.astype(np.float64)
    win = gaussian_window()
    mu1, mu2 = _valid_filter(a, win), _valid_filter(b, win)
    s1 = _valid_filter(a * a, win) - mu1 ** 2
    s2 = _valid_filter(b * b, win) - mu2 ** 2
    s12 = _valid_filter(a * b, win) - mu1 * mu2
    return ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 ** 2 + mu2 ** 2 + C1) * (s1 + s2 + C2))


def calculate_ssim(img1, img2, shave=4):
    if shave and img1.ndim == 3:
        img1 = img1[shave:-shave, shave:-shave, ...]
        img2 = img2[shave:-shave, shave:-shave, ...]
    return float(ssim_map(img1, img2).mean())
