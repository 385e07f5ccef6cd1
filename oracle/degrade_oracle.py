"""numpy restatements of the degradation kernels (TEST INFRASTRUCTURE ONLY).  OpenCV is not installed here, so these
follow OpenCV's / libjpeg's published algorithms (parity with cv2 itself is UNPINNED); the blur-kernel generators are
pinned to the reference's own numpy functions by tests/golden/degrade_kernels.pt (oracle/make_golden_degrade.py)."""
import math

import numpy as np


def filter2d(img, k):
    """cv2.filter2D: correlation, centred anchor, BORDER_REFLECT_101.  img [H,W], k [ks,ks]."""
    r = k.shape[0] // 2
    p = np.pad(img.astype(np.float64), r, mode="reflect")
    out = np.zeros(img.shape, dtype=np.float64)
    for j in range(k.shape[0]):
        for i in range(k.shape[1]):
            out += k[j, i] * p[j:j + img.shape[0], i:i + img.shape[1]]
    return out


def _cubic_w(x):
    A = -0.75
    w0 = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A
    w1 = ((A + 2) * x - (A + 3)) * x * x + 1
    w2 = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1
    return [w0, w1, w2, 1 - w0 - w1 - w2]


def _axis_table(ssize, dsize, mode, area_shrink):
    """per output index: list of (source index, weight)"""
    scale = ssize / dsize
    tab = []
    for d in range(dsize):
        if area_shrink:
            fsx1, fsx2 = d * scale, d * scale + scale
            cell = min(scale, ssize - fsx1)
            sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
            sx2 = min(sx2, ssize - 1)
            sx1 = min(sx1, sx2)
            t = []
            if sx1 - fsx1 > 1e-3:
                t.append((sx1 - 1, (sx1 - fsx1) / cell))
            for sx in range(sx1, sx2):
                t.append((sx, 1.0 / cell))
            if fsx2 - sx2 > 1e-3:
                t.append((sx2, min(min(fsx2 - sx2, 1.0), cell) / cell))
            tab.append(t)
        elif mode == "cubic":
            f = np.float32((d + 0.5) * scale - 0.5)
            b = int(math.floor(f))
            w = _cubic_w(np.float32(f - b))
            tab.append([(min(max(b - 1 + a, 0), ssize - 1), float(w[a])) for a in range(4)])
        else:
            if mode == "area":
                b = int(math.floor(d * scale))
                f = np.float32((d + 1) - (b + 1) * (dsize / ssize))
                f = np.float32(0) if f <= 0 else np.float32(f - math.floor(f))
            else:
                f = np.float32((d + 0.5) * scale - 0.5)
                b = int(math.floor(f))
                f = np.float32(f - b)
            if b < 0:
                b, f = 0, 0.0
            if b >= ssize - 1:
                b, f = ssize - 1, 0.0
            tab.append([(b, 1.0 - float(f)), (min(b + 1, ssize - 1), float(f))])
    return tab


def resize(img, size, mode):
    """cv2.resize for INTER_AREA / INTER_LINEAR / INTER_CUBIC on a float [H,W] plane."""
    H, W = img.shape
    Ho, Wo = size
    shrink = mode == "area" and W / Wo >= 1 and H / Ho >= 1
    tx, ty = _axis_table(W, Wo, mode, shrink), _axis_table(H, Ho, mode, shrink)
    tmp = np.zeros((H, Wo), dtype=np.float64)
    for d, taps in enumerate(tx):
        for (i, w) in taps:
            tmp[:, d] += w * img[:, i]
    out = np.zeros((Ho, Wo), dtype=np.float64)
    for d, taps in enumerate(ty):
        for (i, w) in taps:
            out[d, :] += w * tmp[i, :]
    return out


JQ_LUMA = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87,
                    80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92,
                    95, 98, 112, 100, 103, 99], dtype=np.float64).reshape(8, 8)
JQ_CHROMA = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99,
                      99, 99, 99] + [99] * 32, dtype=np.float64).reshape(8, 8)


def _qtable(base, quality):
    q = min(max(int(quality), 1), 100)
    sf = 5000 // q if q < 50 else 200 - 2 * q
    return np.clip(np.floor((base * sf + 50) / 100), 1, 255)


def _dct_mat():
    k, x = np.mgrid[0:8, 0:8].astype(np.float64)
    m = 0.5 * np.cos((2 * x + 1) * k * math.pi / 16)
    m[0] *= 1 / math.sqrt(2)
    return m


def _quant_plane(p, table):
    D = _dct_mat()
    out = np.empty_like(p)
    for y in range(0, p.shape[0], 8):
        for x in range(0, p.shape[1], 8):
            c = D @ (p[y:y + 8, x:x + 8] - 128.0) @ D.T
            c = np.rint(c / table) * table
            out[y:y + 8, x:x + 8] = D.T @ c @ D + 128.0
    return out


def jpeg(img, quality):
    """img [3,H,W] RGB in [0,1] -> JPEG round trip (4:2:0, float DCT), same staging as csrc/degrade.hip."""
    _, H, W = img.shape
    Hp, Wp = -(-H // 16) * 16, -(-W // 16) * 16
    u8 = np.rint(255 * np.clip(img.astype(np.float32), 0, 1)).astype(np.float32)
    pad = np.pad(u8, ((0, 0), (0, Hp - H), (0, Wp - W)), mode="edge")
    r, g, b = pad.astype(np.float64)
    Y = np.rint(0.299 * r + 0.587 * g + 0.114 * b)
    cb = -0.168735892 * r - 0.331264108 * g + 0.5 * b + 128.0
    cr = 0.5 * r - 0.418687589 * g - 0.081312411 * b + 128.0
    ds = lambda p: np.rint(0.25 * (p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2]))
    Y = _quant_plane(Y, _qtable(JQ_LUMA, quality))
    Cb, Cr = _quant_plane(ds(cb), _qtable(JQ_CHROMA, quality)), _quant_plane(ds(cr), _qtable(JQ_CHROMA, quality))

    def up(p):
        Hc, Wc = p.shape
        ys, xs = np.mgrid[0:Hp, 0:Wp]
        cy, cx = ys >> 1, xs >> 1
        ny = np.clip(cy + np.where(ys & 1, 1, -1), 0, Hc - 1)
        nx = np.clip(cx + np.where(xs & 1, 1, -1), 0, Wc - 1)
        return (9 * p[cy, cx] + 3 * p[cy, nx] + 3 * p[ny, cx] + p[ny, nx]) / 16.0

    yy = np.clip(np.rint(Y), 0, 255)
    cb = np.clip(np.rint(up(Cb)), 0, 255) - 128.0
    cr = np.clip(np.rint(up(Cr)), 0, 255) - 128.0
    rgb = np.stack([yy + 1.402 * cr, yy - 0.344136286 * cb - 0.714136286 * cr, yy + 1.772 * cb])
    return (np.clip(np.rint(rgb), 0, 255) / 255.0)[:, :H, :W]
