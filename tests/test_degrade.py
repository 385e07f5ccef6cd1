"""Real-ESRGAN degradations on the device.  Blur-kernel generators: pinned to the REFERENCE's own
get_gaussian_kernel / get_sinc_kernel (tests/golden/degrade_kernels.pt).  The reference runs the image operations through OpenCV
(dataops/augmentations.py:1666-1801, dataops/augmennt/augmennt/{transforms,functional,extra_functional}.py), which is not installed
here; every device kernel is therefore pinned to an INDEPENDENT installed implementation of the same operation:
  JPEG      libjpeg-turbo through PIL (cv2.imencode / imdecode call the same library with the same defaults: 4:2:0, islow DCT, fancy
            up-sampling) -- BIT FOR BIT, oracle and device kernel alike;
  filter2D  scipy.ndimage.correlate(mode="mirror") (= BORDER_REFLECT_101, centred anchor);
  resize    torch.nn.functional.interpolate for INTER_LINEAR / INTER_CUBIC (half-pixel centres, A = -0.75, clamped borders),
            average pooling for integer INTER_AREA; fractional INTER_AREA by its closed form (coverage-weighted box);
  noise     by its statistics, its structure (grey = equal channels) and its reproducibility."""
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


def pil_jpeg_roundtrip(u8, quality):
    """uint8 RGB [3,H,W] through a real libjpeg(-turbo) encode + decode with OpenCV's defaults (4:2:0, baseline tables)."""
    import io
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(u8.transpose(1, 2, 0)), "RGB").save(buf, format="JPEG", quality=int(quality), subsampling=2)
    buf.seek(0)
    return np.asarray(Image.open(buf).convert("RGB")).transpose(2, 0, 1)


JPEG_SHAPES = ((48, 48), (37, 51), (130, 70), (43, 45), (8, 8), (16, 3), (5, 9), (200, 136))


def test_oracle_jpeg_is_libjpeg_bit_for_bit():
    """oracle/degrade_oracle.jpeg_u8 against libjpeg-turbo (PIL): every pixel equal, for white-noise images (the worst case for
    rounding ties), odd sizes (edge replication in the down-sampled domain, partial MCUs), tiny widths (libjpeg's replication
    fallback) and the whole quality range of the presets (resrgan_noise.yaml: 30..95) plus the extremes."""
    pytest.importorskip("PIL")
    rs = np.random.RandomState(9)
    for shape in JPEG_SHAPES:
        u8 = rs.randint(0, 256, (3,) + shape).astype(np.uint8)
        for q in (5, 30, 50, 75, 90, 95, 100):
            assert np.array_equal(DO.jpeg_u8(u8, q), pil_jpeg_roundtrip(u8, q)), (shape, q)
    smooth = np.rint(255 * np.clip(np.stack([np.outer(np.linspace(0, 1, 64), np.linspace(0.2, 0.9, 80))] * 3) + 0.02 * rs.randn(3, 64, 80), 0, 1))
    assert np.array_equal(DO.jpeg_u8(smooth.astype(np.uint8), 60), pil_jpeg_roundtrip(smooth.astype(np.uint8), 60))


def test_oracle_filter2d_is_scipy_correlate_mirror():
    ndi = pytest.importorskip("scipy.ndimage")
    rs = np.random.RandomState(4)
    img = rs.rand(37, 53)
    for ks in (3, 7, 21):
        k = rs.rand(ks, ks)
        k /= k.sum()
        assert np.abs(DO.filter2d(img, k) - ndi.correlate(img, k, mode="mirror")).max() < 1e-12, ks


def test_oracle_resize_linear_cubic_are_torch_interpolate():
    """INTER_LINEAR / INTER_CUBIC of a float image = half-pixel-centre bilinear / Keys bicubic (A = -0.75) with clamped source
    indices: torch.nn.functional.interpolate(align_corners=False) implements the same definitions independently."""
    import torch.nn.functional as F
    rs = np.random.RandomState(6)
    img = rs.rand(24, 36)
    t = torch.from_numpy(img)[None, None]
    for size in ((20, 31), (61, 80), (15, 22), (24, 36), (50, 13)):
        lin = F.interpolate(t, size=size, mode="bilinear", align_corners=False)[0, 0].numpy()
        cub = F.interpolate(t, size=size, mode="bicubic", align_corners=False)[0, 0].numpy()
        assert np.abs(DO.resize(img, size, "linear") - lin).max() < 5e-6, size       # (the tables carry fp32 coordinates like cv2's)
        assert np.abs(DO.resize(img, size, "cubic") - cub).max() < 5e-6, size


def test_oracle_resize_area_closed_forms():
    img = np.random.RandomState(0).rand(24, 36)
    assert np.allclose(DO.resize(img, (12, 18), "area"), img.reshape(12, 2, 18, 2).mean((1, 3)))      # integer box filter
    # third-party statements of the integer-ratio case (where every definition of "area" agrees): torch's adaptive average
    # pooling (F.interpolate mode='area') and PIL's exact box reduction (Image.reduce), per-axis factors 2, 3, 4, 6 and mixed
    from PIL import Image
    for size in ((12, 18), (8, 12), (6, 9), (4, 6), (12, 9), (6, 36), (24, 12)):
        got = DO.resize(img, size, "area")
        tm = torch.nn.functional.interpolate(torch.from_numpy(img)[None, None], size=size, mode="area")[0, 0].numpy()
        assert np.abs(got - tm).max() < 1e-12, size
        pil = np.asarray(Image.fromarray(img.astype(np.float32), mode="F").reduce((36 // size[1], 24 // size[0])))
        assert np.abs(got - pil).max() < 2e-6, size
    assert np.allclose(DO.resize(img, (24, 36), "linear"), img) and np.allclose(DO.resize(img, (24, 36), "cubic"), img)
    # fractional shrink: every output = coverage-weighted mean of the source cells its box [d s, (d + 1) s) overlaps; the weights of a
    # row sum to 1 and a constant image stays constant
    ones = np.ones((24, 36))
    assert np.allclose(DO.resize(ones, (10, 13), "area"), 1.0) and np.allclose(DO.resize(ones, (7, 36), "area"), 1.0)
    ramp = np.tile(np.arange(36, dtype=np.float64), (24, 1))
    got = DO.resize(ramp, (24, 10), "area")[0]
    # the image is piecewise constant (cell x holds the value x): the box mean is the integral of floor(x) over [3.6 d, 3.6 d + 3.6)
    fine = np.floor((np.arange(36000) + 0.5) / 1000.0)
    want = fine.reshape(10, 3600).mean(1)
    assert np.allclose(got, want, atol=2e-3)       # (cv2's table drops coverage slivers below 1e-3 of a pixel)


@pytest.mark.gpu
def test_device_filter2d_and_resize():
    from trainner_amd.dataops import degradations as D
    rs = np.random.RandomState(2)
    img = rs.rand(2, 3, 37, 53).astype(np.float32)
    ks = [D.gaussian_kernel(21, (2.0, 0.7), angle=30.0), D.sinc_kernel(1.3, 9)]
    got = D.filter2d(torch.from_numpy(img).cuda(), ks).cpu().numpy()
    import scipy.ndimage as ndi
    for n in range(2):
        for c in range(3):
            ref = DO.filter2d(img[n, c], ks[n])
            assert np.abs(got[n, c] - ref).max() < 5e-6, (n, c)
            assert np.abs(got[n, c] - ndi.correlate(img[n, c].astype(np.float64), ks[n], mode="mirror")).max() < 5e-6, (n, c)
    for mode, size in (("area", (10, 13)), ("area", (37, 20)), ("area", (50, 70)), ("area", (18, 53)), ("linear", (20, 31)),
                       ("linear", (61, 80)), ("cubic", (15, 22)), ("cubic", (74, 99)), ("area", (6, 9))):
        got = D.resize(torch.from_numpy(img).cuda(), size, mode).cpu().numpy()
        assert got.shape == (2, 3) + size
        for n in range(2):
            ref = DO.resize(img[n, 1].astype(np.float64), size, mode)
            assert np.abs(got[n, 1] - ref).max() < 2e-5, (mode, size, np.abs(got[n, 1] - ref).max())
        if mode != "area":       # the independent statement of the same definition
            tm = torch.nn.functional.interpolate(torch.from_numpy(img), size=size, mode={"linear": "bilinear", "cubic": "bicubic"}[mode],
                                                 align_corners=False).numpy()
            assert np.abs(got - tm).max() < 2e-5, (mode, size)
    # area at integer ratios (every definition agrees there): torch's adaptive average pooling
    img2 = rs.rand(2, 3, 36, 54).astype(np.float32)
    for size in ((18, 27), (12, 18), (9, 27), (36, 9), (6, 6)):
        got = D.resize(torch.from_numpy(img2).cuda(), size, "area").cpu().numpy()
        tm = torch.nn.functional.interpolate(torch.from_numpy(img2), size=size, mode="area").numpy()
        assert np.abs(got - tm).max() < 2e-6, size


@pytest.mark.gpu
def test_device_jpeg_is_libjpeg_bit_for_bit():
    """tnr_jpeg_sim (integer emulation of libjpeg: csrc/degrade.hip) against the oracle AND against libjpeg-turbo itself (PIL):
    every decoded 8-bit sample equal, batches with per-image qualities, odd sizes, noise images."""
    from trainner_amd.dataops import degradations as D
    rs = np.random.RandomState(3)
    have_pil = True
    try:
        import PIL  # noqa: F401
    except ImportError:
        have_pil = False
    for shape in JPEG_SHAPES:
        u8 = rs.randint(0, 256, (3, 3) + shape).astype(np.uint8)
        q = [30, 77, 95]
        got = D.jpeg(torch.from_numpy(u8.astype(np.float32) / 255.0).cuda(), q).cpu().numpy()
        got8 = np.rint(got * 255).astype(np.uint8)
        assert np.abs(got * 255 - got8).max() < 1e-3                               # decoded samples are 8-bit levels
        for n in range(3):
            assert np.array_equal(got8[n], DO.jpeg_u8(u8[n], q[n])), (shape, n)
            if have_pil:
                assert np.array_equal(got8[n], pil_jpeg_roundtrip(u8[n], q[n])), (shape, n)
    x = rs.rand(2, 3, 40, 40).astype(np.float32)
    y = D.jpeg(torch.from_numpy(np.stack([x[0], x[0]])).cuda(), [95, 30]).cpu().numpy()
    assert np.abs(y[0] - x[0]).mean() < np.abs(y[1] - x[0]).mean()


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


def natural_image(N, H, W, seed):
    """Images with natural statistics: 1/f amplitude spectrum (random phases), per-channel correlated, scaled into [0.05, 0.95]."""
    rs = np.random.RandomState(seed)
    fy, fx = np.fft.fftfreq(H)[:, None], np.fft.fftfreq(W)[None, :]
    amp = 1.0 / np.maximum(np.sqrt(fy * fy + fx * fx), 1.0 / max(H, W))
    out = np.empty((N, 3, H, W), dtype=np.float32)
    for n in range(N):
        base = np.fft.ifft2(amp * np.exp(2j * np.pi * rs.rand(H, W))).real
        for c in range(3):
            ch = base + 0.3 * np.fft.ifft2(amp * np.exp(2j * np.pi * rs.rand(H, W))).real
            ch = (ch - ch.min()) / (ch.max() - ch.min())
            out[n, c] = 0.05 + 0.9 * ch
    return out


@pytest.mark.gpu
def test_composed_pipeline_against_host_libraries():
    """The COMPOSED Real-ESRGAN pipeline (blur -> resize -> noise -> jpeg -> blur2 -> resize2 -> noise2 -> final resize + sinc / jpeg in
    either order: dataops/augmentations.py:1666-1801) on images with natural statistics: one fixed plan per sample, executed by
    RealESRGANDegradation.run on the device and, operation by operation, on the host by the installed libraries -- scipy.ndimage.correlate
    (mirror) for every blur, torch interpolate for linear / cubic resizes (average pooling / the coverage-weighted box for area), PIL's
    libjpeg for every JPEG stage, and for the Gaussian noise stages the device's own draw (a counter-based field: the same seed on a zero
    image, unclipped) added on the host.  Errors compound through 8-bit quantisation at the JPEG stages (a sample within 1e-6 of a
    rounding boundary enters libjpeg one level apart and moves its whole 8 x 8 block): the final LR must agree within ONE 8-bit level on
    >= 99.9 % of the samples, and on average within 0.02 of a level."""
    pytest.importorskip("PIL")
    import scipy.ndimage as ndi
    import torch.nn.functional as F
    from trainner_amd.dataops import degradations as D
    conf = D.degradation_config({"augs_strategy": "resrgan", "lr_noise_types": ["gaussian"], "lr_noise_types2": ["gaussian"]})
    N, H, W = 6, 128, 160
    hr = natural_image(N, H, W, 17)
    deg = D.RealESRGANDegradation(scale=4, preset=conf, seed=23)
    plans = deg.plan(N, H, W)
    kinds = [o[0] for pl in plans for o in pl]
    assert kinds.count("jpeg") >= N and kinds.count("gaussian") >= N and kinds.count("blur") >= 2 * N and kinds.count("resize") >= 2 * N
    assert any(o[0] == "resize" and o[2] == a for pl in plans for o in pl for a in ("area",)) and len({o[2] for pl in plans for o in pl if o[0] == "resize"}) == 3
    got = deg.run(torch.from_numpy(hr).cuda(), plans).cpu().numpy()

    def host_resize(x, size, algo):
        h, w = x.shape[1:]
        if algo == "area":
            if h % size[0] == 0 and w % size[1] == 0:
                return F.avg_pool2d(torch.from_numpy(x)[None], (h // size[0], w // size[1]))[0].numpy()
            return np.stack([DO.resize(x[c], size, "area") for c in range(3)])
        return F.interpolate(torch.from_numpy(x)[None], size=tuple(size), mode={"linear": "bilinear", "cubic": "bicubic"}[algo], align_corners=False)[0].numpy()

    worst = []
    for n, pl in enumerate(plans):
        x = hr[n].astype(np.float64)
        for op in pl:
            if op[0] == "blur":
                x = np.stack([ndi.correlate(x[c], np.asarray(op[1], dtype=np.float64), mode="mirror") for c in range(3)])
            elif op[0] == "resize":
                x = host_resize(x, op[1], op[2])
            elif op[0] == "gaussian":
                field = D.add_gaussian_noise(torch.zeros((1, 3) + x.shape[1:], device="cuda"), [op[1]], [op[2]], op[3], clip=False)[0].cpu().numpy()
                x = np.clip(x + field.astype(np.float64), 0.0, 1.0)
            else:
                assert op[0] == "jpeg"
                u8 = np.rint(np.clip(x, 0.0, 1.0) * 255.0).astype(np.uint8)
                x = pil_jpeg_roundtrip(u8, op[1]).astype(np.float64) / 255.0
        x = np.clip(x, 0.0, 1.0)
        assert x.shape == (3, H // 4, W // 4)
        d = np.abs(got[n].astype(np.float64) - x) * 255.0
        worst.append((float((d <= 1.0).mean()), float(d.mean()), float(d.max())))
    frac = np.mean([w[0] for w in worst])
    assert frac >= 0.999 and np.mean([w[1] for w in worst]) <= 0.02, worst


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


def _grammar(ops, lr_size):
    """Split a planned operation list by the control flow of aug_pipeline (dataops/augmentations.py:1666-1801); returns the facts the
    statistics below are taken over, or raises AssertionError when the list is not a sentence of that grammar."""
    i, facts = 0, {}

    def take(kind):
        nonlocal i
        if i < len(ops) and ops[i][0] == kind:
            i += 1
            return ops[i - 1]
        return None

    def noise():
        nonlocal i
        assert i < len(ops) and ops[i][0] in ("gaussian", "poisson"), ops[i:i + 1]
        i += 1
        return ops[i - 1]

    facts["blur1"], facts["resize1"], facts["noise1"] = take("blur"), take("resize"), noise()
    facts["jpeg1"] = take("jpeg")
    assert facts["jpeg1"] is not None
    facts["blur2"], facts["resize2"], facts["noise2"] = take("blur"), take("resize"), noise()
    # the second resize and the final resize are both ("resize", ...): the final one targets the LR size and is followed by blur / jpeg / end
    tail = ops[i:]
    kinds = [o[0] for o in tail]
    assert kinds in (["jpeg", "resize"], ["jpeg", "resize", "blur"], ["resize", "jpeg"], ["resize", "blur", "jpeg"]), kinds
    facts["compress_first"] = kinds[0] == "jpeg"
    facts["final_sinc"] = "blur" in kinds
    fr = [o for o in tail if o[0] == "resize"][0]
    assert tuple(fr[1]) == tuple(lr_size)
    facts["final_algo"], facts["jpeg2"] = fr[2], [o for o in tail if o[0] == "jpeg"][0]
    return facts


def test_plan_replays_the_reference_control_flow():
    """The host half of the pipeline (RealESRGANDegradation.plan: every random decision, no device) against aug_pipeline's control
    flow and the resrgan presets' probabilities / ranges: 1 500 planned samples must all be sentences of
        [blur] [resize] noise jpeg [blur2] [resize2] noise2 ( jpeg resize [sinc] | resize [sinc] jpeg )
    with blur 100 %, blur2 80 %, sinc 80 %, either final order 50 %, resize kept 10 % / 30 %, noise kinds 50 / 50, grey 40 %,
    qualities in 30..95, odd kernel sizes 7..21, the three resize algorithms equally likely."""
    from trainner_amd.dataops.degradations import RealESRGANDegradation
    d = RealESRGANDegradation(scale=4, seed=123)
    H = W = 256
    F = [_grammar(ops, (H // 4, W // 4)) for _ in range(30) for ops in d.plan(50, H, W)]
    n = float(len(F))

    def frac(pred):
        return sum(1 for f in F if pred(f)) / n

    tol = 4.0 * 0.5 / math.sqrt(n)          # 4 sigma of a fair coin: generous for every proportion below
    assert frac(lambda f: f["blur1"] is not None) == 1.0
    assert abs(frac(lambda f: f["blur2"] is not None) - 0.8) < tol and abs(frac(lambda f: f["final_sinc"]) - 0.8) < tol
    assert abs(frac(lambda f: f["compress_first"]) - 0.5) < tol
    assert abs(frac(lambda f: f["resize1"] is None) - 0.1) < tol and abs(frac(lambda f: f["resize2"] is None) - 0.3) < tol
    assert abs(frac(lambda f: f["noise1"][0] == "gaussian") - 0.5) < tol and abs(frac(lambda f: f["noise2"][0] == "poisson") - 0.5) < tol
    assert abs(frac(lambda f: f["noise1"][2]) - 0.4) < tol                                  # grey = 1 - prob_color
    for a in ("area", "linear", "cubic"):
        assert abs(frac(lambda f: f["final_algo"] == a) - 1 / 3) < tol
    qs = [f["jpeg1"][1] for f in F] + [f["jpeg2"][1] for f in F]
    assert min(qs) == 30 and max(qs) == 95
    ks = [f["blur1"][1].shape[0] for f in F]
    assert all(k % 2 == 1 and 7 <= k <= 21 for k in ks) and min(ks) == 7 and max(ks) == 21
    f1 = [f["resize1"][1][0] / H for f in F if f["resize1"] is not None]
    assert min(f1) >= 0.15 - 1 / H and max(f1) <= 1.5 + 1 / H and abs(np.mean([x > 1 for x in f1]) - 0.2 / 0.9) < tol
    g = [f["noise1"][1] for f in F if f["noise1"][0] == "gaussian"]
    assert all(1.0 <= s ** 2 <= 30.0 + 1e-9 for sig in g for s in sig)
    multi = np.mean([len(set(sig)) == 3 for sig in g])                                      # MC-AWGN: 34 % of the colour draws
    assert abs(multi - 0.34 * 0.6) < tol
    ps = [f["noise2"][1] for f in F if f["noise2"][0] == "poisson"]
    assert min(ps) >= 0.05 and max(ps) <= 2.5
    # same seed, same plan; the noise seeds differ per sample and per call
    a, b = RealESRGANDegradation(scale=4, seed=9).plan(3, 64, 64), RealESRGANDegradation(scale=4, seed=9).plan(3, 64, 64)
    assert str(a) == str(b)
    seeds = [op[3] for ops in a for op in ops if op[0] in ("gaussian", "poisson")]
    assert len(set(seeds)) == len(seeds)


def test_presets_are_read_and_overridden(tmp_path):
    """degradation_config = the reference's preset overlay (options/options.py:148-165,366-463): built-in values == the reference's
    own resrgan_*.yaml files; a user-edited preset, `add_*_preset`, and keys in the dataset options change the draws."""
    import yaml
    from trainner_amd.dataops.degradations import RESRGAN, RealESRGANDegradation, degradation_config
    ref_root = "/root/reference/codes/options/presets"
    if os.path.isdir(ref_root):           # build container: the constants ARE the reference's files
        assert degradation_config({"augs_strategy": "resrgan"}, presets_root=ref_root) == degradation_config() == RESRGAN
    root = tmp_path / "presets"
    root.mkdir()
    (root / "mine_blur.yaml").write_text(yaml.safe_dump({"kind": "Blur", "config": {
        "pipeline": {"lr_blur": True, "lr_blur_types": {"iso": 1.0}, "blur_prob": 1.0, "lr_blur2": False, "final_blur": ["sinc"], "final_blur_prob": 0.0},
        "iso": {"min_kernel_size": 9, "kernel_size": 9, "sigmaX": [1.0, 1.0]}}}))
    (root / "mine_noise.yaml").write_text(yaml.safe_dump({"kind": "Noise", "config": {
        "pipeline": {"lr_noise": True, "lr_noise_types": ["gaussian"], "lr_noise2": True, "lr_noise_types2": ["poisson"], "compression": ["jpeg"],
                     "final_compression": ["jpeg"]},
        "gaussian": {"var_limit": [4, 4], "prob_color": 1.0, "multi": False}, "jpeg": {"min_quality": 50, "max_quality": 50}}}))
    ds = {"augs_strategy": "resrgan", "add_blur_preset": "mine_blur", "add_noise_preset": "mine_noise", "lr_downscale2": False}
    conf = degradation_config(ds, presets_root=str(root))
    assert conf["blur"]["types"] == {"iso": 1.0} and conf["blur2"]["enabled"] is False and conf["resize2"]["enabled"] is False
    assert conf["resize"] == RESRGAN["resize"]                                             # untouched stage keeps the built-in values
    for ops in RealESRGANDegradation(scale=4, preset=conf, seed=1).plan(40, 64, 64):
        kinds = [o[0] for o in ops]
        assert kinds.count("blur") == 1 and ops[0][0] == "blur" and ops[0][1].shape == (9, 9)     # one iso blur of 9 taps, no blur2, no sinc
        assert [o for o in ops if o[0] == "gaussian"][0][1] == [2.0, 2.0, 2.0] and not [o for o in ops if o[0] == "gaussian"][0][2]
        assert "poisson" in kinds and all(o[1] == 50 for o in ops if o[0] == "jpeg")
        assert kinds.count("resize") <= 2
    with pytest.raises(NotImplementedError):
        degradation_config({"augs_strategy": "resrgan", "lr_noise_types": ["camera"]})
    # everything starts from the reference's base presets (stages OFF, plain down-scaling on): a single preset switches on
    # ITS stages only; a missing preset is skipped like in the reference (`realsr` ships no blur preset); resize / blur / noise types
    # the device pipeline lacks raise NotImplementedError (never a KeyError)
    from trainner_amd.dataops.degradations import BASE
    only_blur = degradation_config({"add_blur_preset": "resrgan_blur"})
    assert only_blur["blur"] == RESRGAN["blur"] and only_blur["blur2"] == RESRGAN["blur2"] and only_blur["final_blur"] == RESRGAN["final_blur"]
    for k in ("noise", "noise2", "compression", "final_compression", "resize", "resize2"):
        assert only_blur[k]["enabled"] is False, k
    assert only_blur["final_scale"] == BASE["final_scale"] and only_blur["final_scale"]["enabled"] is True
    for ops in RealESRGANDegradation(scale=4, preset=only_blur, seed=3).plan(20, 64, 64):
        assert {o[0] for o in ops} <= {"blur", "resize"} and [tuple(o[1]) for o in ops if o[0] == "resize"] == [(16, 16)]
    assert degradation_config({"augs_strategy": "bsrgan"}, presets_root=str(root)) == BASE          # no such files: skipped
    if os.path.isdir(ref_root):
        for strat in ("realsr", "bsrgan", "combo"):      # (realsr: its missing blur preset is skipped; its `realistic` resize is what raises)
            with pytest.raises(NotImplementedError):
                degradation_config({"augs_strategy": strat}, presets_root=ref_root)
        assert degradation_config({"add_blur_preset": "resrgan_blur"}, presets_root=ref_root) == only_blur
    # options.parse carries the merged configuration into the train dataset options
    from oracle import ref_harness
    from trainner_amd.options import options
    yml = ref_harness.esrgan_yaml(name="presets", out_root=str(tmp_path), gpu_ids="[0]", nb=1, batch=2, crop=64, d_nf=16)
    txt = open(yml).read().replace("    dataroot_LR: /tmp/none_lr\n", "    augs_strategy: resrgan\n    add_blur_preset: mine_blur\n")
    txt = txt.replace("use_tb_logger: false", "use_tb_logger: false\npresets_root: %s" % root)
    open(yml, "w").write(txt)
    opt = options.parse(yml, is_train=True)
    dsopt = opt["datasets"]["train"]
    assert dsopt["degradation"]["blur"]["types"] == {"iso": 1.0} and dsopt["presets_root"] == str(root)


@pytest.mark.gpu
def test_batched_leading_stages_equal_per_sample_execution():
    """run(): the operations all samples share at the head of their plans (the full-size first blur, always) go out as one launch with
    per-sample parameters -- bit-identical to executing every sample on its own."""
    from trainner_amd.dataops.degradations import RealESRGANDegradation
    hr = torch.rand(6, 3, 96, 128, generator=torch.Generator().manual_seed(2)).cuda()
    d = RealESRGANDegradation(scale=4, seed=11)
    plans = d.plan(6, 96, 128)
    assert all(p[0][0] == "blur" for p in plans)
    whole = d.run(hr, plans)
    for n in range(6):
        assert torch.equal(whole[n:n + 1], d.run(hr[n:n + 1], [plans[n]])), n
