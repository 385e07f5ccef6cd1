"""Patch-wise inference helpers of the validation path (reference: codes/dataops/common.py:575-767), the two
functions SRModel.test_chop() needs.  Behaviour follows the reference exactly, quirks included, and is pinned by
reference-generated fixtures (tests/golden/patches.pt, oracle/make_golden_patches.py)."""
import torch
import torch.nn.functional as F


def _window_starts(size, patch, step):
    """Start offsets of sliding windows: 0, step, 2*step, ... plus a final window flush with the border when the
    regular grid does not end there (dataops/common.py:611-619)."""
    starts = list(range(0, size - patch + 1, step))
    if (size - patch) % step != 0:
        starts.append(size - patch)
    return starts


def extract_patches_2d(img, patch_shape, step=None, batch_first=False):
    """[B,C,H,W] -> [n_patches, B, C, pH, pW] (or [B, n_patches, ...] with batch_first), rows first.
    A float step is relative to the patch size; images smaller than the patch are zero-padded around the centre."""
    step = [1.0, 1.0] if step is None else step
    pH, pW = patch_shape
    if img.size(2) < pH:
        top = (pH - img.size(2)) // 2
        img = F.pad(img, (0, 0, top, pH - img.size(2) - top))
    if img.size(3) < pW:
        left = (pW - img.size(3)) // 2
        img = F.pad(img, (left, pW - img.size(3) - left, 0, 0))
    sH = int(pH * step[0]) if isinstance(step[0], float) else step[0]
    sW = int(pW * step[1]) if isinstance(step[1], float) else step[1]
    rows = _window_starts(img.size(2), pH, sH)
    cols = _window_starts(img.size(3), pW, sW)
    patches = torch.stack([img[:, :, r:r + pH, c:c + pW] for r in rows for c in cols], 0)
    return patches.permute(1, 0, 2, 3, 4) if batch_first else patches


def recompose_tensor(patches, height, width, step=None, scale=1):
    """Blend super-resolved square patches [n, C, P, P] (P = scale * LR patch, row-major over the patch grid of
    extract_patches_2d) back into [B, C, scale*height, scale*width].  Overlaps (step in [0.5, 1.0]) are
    cross-faded with a 0.1 -> 1.0 linear ramp; every patch is placed at min(i * int(step * P), full - P)."""
    step = 1.0 if step is None else step
    assert isinstance(step, float) and 0.5 <= step <= 1.0
    full_h, full_w = scale * height, scale * width
    n, channels, P, _ = patches.shape
    overlap = scale * int(round((1.0 - step) * (P / scale)))
    eff = int(step * P)
    img_h, img_w = max(full_h, P), max(full_w, P)
    s_int = int(P * step)
    n_h = 1 + (img_h - P) // s_int + (1 if (img_h - P) % s_int != 0 else 0)
    n_w = 1 + (img_w - P) // s_int + (1 if (img_w - P) % s_int != 0 else 0)
    batch = n // (n_h * n_w)
    dev, dt = patches.device, patches.dtype
    profile = torch.cat([torch.linspace(0.1, 1.0, overlap), torch.ones(P - 2 * overlap), torch.linspace(1.0, 0.1, overlap)])
    blend = (profile[None, :] * profile[:, None]).to(device=dev, dtype=dt)
    weight = torch.zeros(1, channels, full_h, full_w, device=dev, dtype=dt)
    out = torch.zeros(batch, channels, full_h, full_w, device=dev, dtype=dt)
    idx = 0
    for b in range(batch):
        for h in range(n_h):
            y0 = min(h * eff, full_h - P)
            for w in range(n_w):
                x0 = min(w * eff, full_w - P)
                if b == 0:
                    weight[0, :, y0:y0 + P, x0:x0 + P] += blend
                out[b, :, y0:y0 + P, x0:x0 + P] += patches[idx] * blend
                idx += 1
    return out / weight


def tensor2np(img, rgb2bgr=True, remove_batch=True, data_range=255, denormalize=False, change_range=True, imtype=None):
    """Network output -> uint8 image(s), on the device (reference: codes/dataops/common.py:502-566; used by
    train.py:324-363 before the metrics and image saving).  Arithmetic is the reference's: optional (x + 1) / 2,
    round_half_even(clip(255 x, 0, 255)), RGB -> BGR for 3 / 4-channel tensors, CHW -> HWC.
    Returns a CUDA uint8 tensor: [H,W,C] for a 3-D input or `remove_batch` (image 0 of the batch, as the reference
    keeps), else [N,H,W,C] (the reference builds a make_grid mosaic there; the engine keeps the batch so that
    utils.metrics can score every image in one launch).  `.cpu().numpy()` gives the reference's array."""
    from .. import hip
    if not torch.is_tensor(img):
        raise TypeError("Got unexpected object type, expected torch.Tensor")
    if data_range != 255 or not change_range or imtype not in (None, "uint8"):
        raise NotImplementedError("tensor2np on the HIP engine produces uint8 images in [0, 255]")
    hip.require_device(img)
    t = img.detach().float()
    if t.dim() == 2:
        t = t[None, None]
        squeeze = "hw"
    elif t.dim() == 3:
        t = t[None]
        squeeze = "hwc"
    elif t.dim() == 4:
        if remove_batch:
            t = t[:1]
        squeeze = "hwc" if remove_batch else None
    else:
        raise TypeError("Only support 4D, 3D and 2D tensor. But received with dimension: {:d}".format(t.dim()))
    t = t.contiguous()
    N, C, H, W = t.shape
    out = torch.empty((N, H, W, C), dtype=torch.uint8, device=t.device)
    hip.check(hip.load().tnr_tensor2np_u8(t.data_ptr(), N, C, H, W, out.data_ptr(), int(bool(rgb2bgr)), int(bool(denormalize)),
                                         hip.stream()), "tensor2np_u8")
    if squeeze == "hw":
        return out[0, :, :, 0]
    if squeeze == "hwc":
        return out[0]
    return out


def np2tensor(img, bgr2rgb=True, data_range=1.0, normalize=False, change_range=True, add_batch=True):
    """uint8 HWC image (numpy array, or uint8 tensor) -> fp32 tensor on the device (reference: codes/dataops/common.py:
    470-499): x * data_range / 255, HWC -> CHW, BGR(A) -> RGB(A), optional norm(); [1,C,H,W] with add_batch."""
    import numpy as np
    from .. import hip
    if isinstance(img, np.ndarray):
        img = torch.from_numpy(np.ascontiguousarray(img))
    if not torch.is_tensor(img) or img.dtype != torch.uint8 or not change_range:
        raise TypeError("np2tensor on the HIP engine converts uint8 images")
    if img.dim() == 2:
        img = img[:, :, None]
    hip.require_device()
    src = img.to("cuda").contiguous()
    H, W, C = src.shape
    out = torch.empty((1, C, H, W), dtype=torch.float32, device=src.device)
    hip.check(hip.load().tnr_feed_u8_to_tensor(src.data_ptr(), 1, H, W, C, None, 0, out.data_ptr(), int(bool(bgr2rgb)),
                                              float(data_range), int(bool(normalize)), hip.stream()), "feed_u8_to_tensor")
    return out if add_batch else out[0]
