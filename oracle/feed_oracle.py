"""numpy restatement of the reference's per-sample input path (TEST INFRASTRUCTURE ONLY):
crop (dataops/augmentations.py:776-790) -> flip (:793-802) -> rotate90 (:805-830) -> np2tensor (dataops/common.py:
470-499).  Pinned by tests/golden/feed.pt, which oracle/make_golden_feed.py produces by calling those reference
functions themselves."""
import numpy as np


def crop(img, pos, size):
    x1, y1 = pos
    oh, ow = img.shape[:2]
    if ow > size or oh > size:
        return img[y1:y1 + size, x1:x1 + size, ...]
    return img


def flip_rot(img, flip, rot, vflip):
    if flip:
        img = np.flip(img, axis=1)
    if rot:
        if vflip:
            img = np.flip(img, axis=0)
        img = np.rot90(img, 1)
    return img


def np2tensor(img, bgr2rgb=True, data_range=1.0, normalize=False):
    """-> float32 [C,H,W]"""
    x = img * data_range / 255                                   # uint8 * float -> float64
    t = np.ascontiguousarray(np.transpose(x, (2, 0, 1))).astype(np.float32)
    if bgr2rgb:
        if t.shape[0] % 3 == 0:
            t = t[::-1].copy()
        elif t.shape[0] == 4:
            t = t[[2, 1, 0, 3]].copy()
    if normalize:
        t = np.clip((t - np.float32(0.5)) * np.float32(2.0), -1, 1)
    return t


def flags_of(flip, rot, vflip):
    f = (1 if flip else 0) | (2 if rot else 0)
    return f | (4 if (rot and vflip) else 0)
