"""Real-ESRGAN degradations on the device.  Blur-kernel generators: pinned to the REFERENCE's own
get_gaussian_kernel / get_sinc_kernel (tests/golden/degrade_kernels.pt).  Device kernels: against the numpy
restatements of OpenCV's / libjpeg's published algorithms in oracle/degrade_oracle.py (cv2 is not installed: parity
with cv2 itself is unpinned) -- filter2D / resize to fp32 round-off, JPEG exactly up to rounding ties, noise by its
statistics, its structure (grey = equal channels) and its reproducibility."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import degrade_oracle as DO

KFX = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "degrade_kernels.pt"), weights_only=False)


def test_blur_kernel_generators_match_reference():
    from trainner_amd.dataops import degradations as D
    for c in KFX["gauss"]:
        k = D.gaussian_kernel(c["ks"], c["sigma"])
        assert k.shape == (c["ks"], c["ks"]) and np.allclose(k, c["kernel"].numpy(), rtol=1e-6, atol=1e-12), c["ks"]
    for c in KFX["sinc"]:
        k = D.sinc_kernel(c["cutoff"], c["ks"])
        assert np.allclose(k, c["kernel"].numpy(), rtol=1e-6, atol=1e-12), c["ks"]
    # rotation (cv2.getRotationMatrix2D + warpAffine restated): 0 deg = identity, 90 deg swaps the axes, sum stays 1
    base = D.gaussian_kernel(21, (0.6, 2.5))
    assert np.allclose(D.rotate_kernel(base, 0.0), base)
    r90 = D.gaussian_kernel(21, (0.6, 2.5), angle=90.0)
    assert np.allclose(r90, D.gaussian_kernel(21, (2.5, 0.6)), atol=1e-9) and abs(r90.sum() - 1) < 1e-12
    assert D.pad_kernel21(np.ones((7, 7))).sum() == 49 and D.pad_kernel21(np.ones((7, 7)))[7:14, 7:14].all()


def test_oracle_resize_and_jpeg_sanity():
    img = np.random.RandomState(0).rand(24, 36)
    assert np.allclose(DO.resize(img, (12, 18), "area"), img.reshape(12, 2, 18, 2).mean((1, 3)))      # integer box filter
    assert np.allclose(DO.resize(img, (24, 36), "linear"), img) and np.allclose(DO.resize(img, (24, 36), "cubic"), img)
    flat = np.full((3, 32, 48), 0.5)
    assert np.abs(DO.jpeg(flat, 30) - np.rint(127.5) / 255).max() < 2 / 255      # a flat image survives any quality
    x = np.random.RandomState(1).rand(3, 40, 40)
    assert np.abs(DO.jpeg(x, 95) - x).mean() < np.abs(DO.jpeg(x, 30) - x).mean()


@pytest.mark.gpu
def test_device_filter2d_and_resize():
    from trainner_amd.dataops import degradations as D
    rs = np.random.RandomState(2)
    img = rs.rand(2, 3, 37, 53).astype(np.float32)
    ks = [D.gaussian_kernel(21, (2.0, 0.7), angle=30.0), D.sinc_kernel(1.3, 9)]
    got = D.filter2d(torch.from_numpy(img).cuda(), ks).cpu().numpy()
    for n in range(2):
        for c in range(3):
            ref = DO.filter2d(img[n, c], ks[n])
            assert np.abs(got[n, c] - ref).max() < 5e-6, (n, c)
    for mode, size in (("area", (10, 13)), ("area", (37, 20)), ("area", (50, 70)), ("area", (18, 53)), ("linear", (20, 31)),
                       ("linear", (61, 80)), ("cubic", (15, 22)), ("cubic", (74, 99)), ("area", (6, 9))):
        got = D.resize(torch.from_numpy(img).cuda(), size, mode).cpu().numpy()
        assert got.shape == (2, 3) + size
        for n in range(2):
            ref = DO.resize(img[n, 1].astype(np.float64), size, mode)
            assert np.abs(got[n, 1] - ref).max() < 2e-5, (mode, size, np.abs(got[n, 1] - ref).max())


@pytest.mark.gpu
def test_device_jpeg_simulation():
    from trainner_amd.dataops import degradations as D
    rs = np.random.RandomState(3)
    base = rs.rand(2, 3, 8, 8).astype(np.float32).repeat(6, 2).repeat(6, 3)[:, :, :43, :45]   # blocky image + odd size
    img = np.clip(base + 0.05 * rs.randn(*base.shape).astype(np.float32), 0, 1)
    q = [35, 90]
    got = D.jpeg(torch.from_numpy(img.copy()).cuda(), q).cpu().numpy()
    for n in range(2):
        ref = DO.jpeg(img[n], q[n])
        d = np.abs(got[n] - ref) * 255
        # float DCT on both sides (fp32 on the device, fp64 here): identical except where a coefficient or a decoded
        # sample sits within round-off of a rounding boundary -- one quantisation step / one 8-bit level there
        assert (d > 0.5).mean() < 0.08 and np.median(d) < 1e-3 and d.max() <= 24, (n, (d > 0.5).mean(), d.max())
        assert np.abs(got[n] * 255 - np.rint(got[n] * 255)).max() < 1e-3          # decoded samples are 8-bit levels
    assert np.abs(got[1] - img[1]).mean() < np.abs(got[0] - img[0]).mean() + 0.02


@pytest.mark.gpu
def test_device_noise_statistics_and_reproducibility():
    from trainner_amd.dataops import degradations as D
    x0 = torch.full((3, 3, 128, 160), 0.5, device="cuda")
    a = D.add_gaussian_noise(x0.clone(), [[10.0, 20.0, 5.0], [15.0] * 3, [8.0] * 3], [False, True, False], seed=7, clip=False)
    b = D.add_gaussian_noise(x0.clone(), [[10.0, 20.0, 5.0], [15.0] * 3, [8.0] * 3], [False, True, False], seed=7, clip=False)
    c = D.add_gaussian_noise(x0.clone(), [[10.0, 20.0, 5.0], [15.0] * 3, [8.0] * 3], [False, True, False], seed=8, clip=False)
    assert torch.equal(a, b) and not torch.equal(a, c)                              # counter-based: seed decides everything
    d = (a - 0.5).cpu().numpy() * 255
    for ch, s in enumerate((10.0, 20.0, 5.0)):
        assert abs(d[0, ch].std() - s) < 0.03 * s + 0.05 and abs(d[0, ch].mean()) < 4 * s / math.sqrt(128 * 160)
    assert np.array_equal(d[1, 0], d[1, 1]) and np.array_equal(d[1, 0], d[1, 2]) and abs(d[1, 0].std() - 15.0) < 0.5   # grey
    assert abs(np.corrcoef(d[0, 0].ravel(), d[0, 1].ravel())[0, 1]) < 0.02          # colour noise: independent channels
    # Poisson: variance of Poisson(x * vals) / vals is x / vals; scale multiplies the deviation
    lev = torch.tensor([0.2, 0.5, 0.6], device="cuda").view(1, 3, 1, 1).expand(2, 3, 128, 160).contiguous()
    p = D.add_poisson_noise(lev.clone(), [1.0, 2.0], [False, True], seed=11, vals=[64.0, 256.0], clip=False)
    dp = (p - lev).cpu().numpy()
    for ch, xv in enumerate((0.2, 0.5, 0.6)):          # (0.6: the clip of Poisson / vals to [0, 1] stays 4 sigma away)
        assert abs(dp[0, ch].var() - xv / 64.0) < 0.06 * xv / 64.0 and abs(dp[0, ch].mean()) < 5e-3
    assert np.allclose(dp[1, 0], dp[1, 1], atol=1e-6) and np.allclose(dp[1, 0], dp[1, 2], atol=1e-6)   # grey: the luma deviation on all channels
    want = 4.0 * (0.299 ** 2 * 0.2 + 0.587 ** 2 * 0.5 + 0.114 ** 2 * 0.6) / 256.0
    assert abs(dp[1, 0].var() - want) < 0.08 * want
    assert D.poisson_levels(torch.tensor([[[[0.0, 1.0 / 255, 2.0 / 255, 0.5]]]], device="cuda")) == [4.0]


@pytest.mark.gpu
def test_realesrgan_pipeline_shapes_and_determinism():
    from trainner_amd.dataops.degradations import RealESRGANDegradation
    hr = torch.rand(4, 3, 128, 128, generator=torch.Generator().manual_seed(0)).cuda()
    a = RealESRGANDegradation(scale=4, seed=5)(hr)
    b = RealESRGANDegradation(scale=4, seed=5)(hr)
    c = RealESRGANDegradation(scale=4, seed=6)(hr)
    assert tuple(a.shape) == (4, 3, 32, 32) and a.is_cuda and float(a.min()) >= 0 and float(a.max()) <= 1
    assert torch.equal(a, b) and not torch.equal(a, c)
    # the LR image still carries the HR content: correlates with the area-downscaled HR far better than with noise
    from trainner_amd.dataops import degradations as D
    ref = D.resize(hr, (32, 32), "area")
    smooth = torch.nn.functional.avg_pool2d(hr, 16).repeat_interleave(16, 2).repeat_interleave(16, 3)    # low-frequency test image
    lo = RealESRGANDegradation(scale=4, seed=9)(smooth)
    assert float((lo - D.resize(smooth, (32, 32), "area")).abs().mean()) < 0.12
    assert torch.isfinite(a).all() and ref.shape == a.shape
