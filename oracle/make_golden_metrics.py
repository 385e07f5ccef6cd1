"""tests/golden/metrics.pt from the REFERENCE's own tensor2np (dataops/common.py:502-566), calculate_psnr (utils/metrics.py:110-126)
and calculate_ssim (utils/metrics.py:180-223) -- run in the build container:  python -m oracle.make_golden_metrics
cv2 is not installed: the reference's SSIM code runs with its two OpenCV calls (getGaussianKernel, filter2D) served by scipy
(oracle/stubs/cv2: scipy.signal.windows.gaussian, scipy.ndimage.correlate(mode="mirror") = BORDER_REFLECT_101) -- an implementation
independent of oracle/metrics_oracle.py and of csrc/metrics.hip, which both restate the 'valid' correlation directly.
"""
import os

import torch

from . import detrand
from . import ref_harness as R

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "metrics.pt")


def ssim_value(v):
    """calculate_ssim returns the MAP (not its mean) for [H, W, 1] inputs (utils/metrics.py:219-220): the fixture keeps its mean."""
    import numpy as np
    return float(np.mean(v))


def main():
    cases = {}
    with R.reference_env():
        from dataops.common import tensor2np
        from utils.metrics import calculate_psnr, calculate_ssim
        specs = {"rgb_unit": ((2, 3, 40, 56), 0.0, 1.0, False), "rgb_overshoot": ((1, 3, 33, 47), -0.3, 1.3, False),
                 "rgb_znorm": ((1, 3, 24, 24), -1.2, 1.2, True), "gray": ((1, 1, 30, 30), 0.0, 1.0, False)}
        for name, (shape, lo, hi, den) in specs.items():
            sr = detrand.uniform(shape, 500 + len(cases), lo, hi)
            hr = (sr + detrand.uniform(shape, 600 + len(cases), -0.05, 0.05)).clamp(lo, hi)
            # values on exact .5 boundaries of the 255 grid exercise round-half-even
            sr.view(-1)[:8] = torch.tensor([0.5, 1.5, 2.5, 3.5, 126.5, 127.5, 253.5, 254.5]) / 255.0 * (2.0 if den else 1.0) - (1.0 if den else 0.0)
            imgs = []
            for n in range(shape[0]):
                a = tensor2np(sr[n], denormalize=den)
                b = tensor2np(hr[n], denormalize=den)
                imgs.append(dict(sr_u8=torch.from_numpy(a.copy()), hr_u8=torch.from_numpy(b.copy()),
                                 psnr4=calculate_psnr(a, b, 4), psnr0=calculate_psnr(a, b, 0),
                                 ssim4=ssim_value(calculate_ssim(a, b, 4)), ssim0=ssim_value(calculate_ssim(a, b, 0))))
            cases[name] = dict(sr=sr, hr=hr, denormalize=den, images=imgs)
    torch.save(cases, OUT)
    print("wrote", OUT, {k: [round(i["psnr4"], 4) for i in v["images"]] for k, v in cases.items()})


if __name__ == "__main__":
    main()
