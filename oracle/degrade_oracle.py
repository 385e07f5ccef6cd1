"""numpy restatements of the degradation kernels (TEST INFRASTRUCTURE ONLY).  OpenCV is not installed here, so these follow the
algorithms of the libraries the reference calls through OpenCV, and each is pinned to an INDEPENDENT implementation that is installed
(tests/test_degrade.py): the JPEG round trip bit for bit to libjpeg-turbo through PIL (cv2.imencode / imdecode run the same library),
filter2D to scipy.ndimage.correlate(mode="mirror") (= BORDER_REFLECT_101), the resize tables to closed forms; the blur-kernel
generators are pinned to the reference's own numpy functions by tests/golden/degrade_kernels.pt (oracle/make_golden_degrade.py)."""
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
                    95, 98, 112, 100, 103, 99], dtype=np.int64).reshape(8, 8)
JQ_CHROMA = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99,
                      99, 99, 99] + [99] * 32, dtype=np.int64).reshape(8, 8)


def _qtable(base, quality):
    """libjpeg jpeg_quality_scaling + jpeg_add_quant_table (force_baseline)."""
    q = min(max(int(quality), 1), 100)
    sf = 5000 // q if q < 50 else 200 - 2 * q
    return np.clip((base.astype(np.int64) * sf + 50) // 100, 1, 255)


# ---- libjpeg's "islow" integer DCT (jfdctint.c / jidctint.c): 13-bit constants, two 8-point passes, PASS1_BITS = 2
_CB, _P1 = 13, 2
_F = dict(f0_298=2446, f0_390=3196, f0_541=4433, f0_765=6270, f0_899=7373, f1_175=9633, f1_501=12299, f1_847=15137, f1_961=16069,
          f2_053=16819, f2_562=20995, f3_072=25172)


def _ds(x, n):
    return (x + (1 << (n - 1))) >> n


def _fdct8(d, first):
    """8-point forward pass along the last axis (int64)."""
    d0, d1, d2, d3, d4, d5, d6, d7 = [d[..., i] for i in range(8)]
    t0, t7, t1, t6, t2, t5, t3, t4 = d0 + d7, d0 - d7, d1 + d6, d1 - d6, d2 + d5, d2 - d5, d3 + d4, d3 - d4
    t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    o = [None] * 8
    sh = _CB - _P1 if first else _CB + _P1
    o[0] = (t10 + t11) << _P1 if first else _ds(t10 + t11, _P1)
    o[4] = (t10 - t11) << _P1 if first else _ds(t10 - t11, _P1)
    z1 = (t12 + t13) * _F["f0_541"]
    o[2] = _ds(z1 + t13 * _F["f0_765"], sh)
    o[6] = _ds(z1 - t12 * _F["f1_847"], sh)
    z1, z2, z3, z4 = t4 + t7, t5 + t6, t4 + t6, t5 + t7
    z5 = (z3 + z4) * _F["f1_175"]
    t4, t5, t6, t7 = t4 * _F["f0_298"], t5 * _F["f2_053"], t6 * _F["f3_072"], t7 * _F["f1_501"]
    z1, z2, z3, z4 = -z1 * _F["f0_899"], -z2 * _F["f2_562"], -z3 * _F["f1_961"] + z5, -z4 * _F["f0_390"] + z5
    o[7], o[5], o[3], o[1] = _ds(t4 + z1 + z3, sh), _ds(t5 + z2 + z4, sh), _ds(t6 + z2 + z3, sh), _ds(t7 + z1 + z4, sh)
    return np.stack(o, -1)


def _idct8(x, first):
    i0, i1, i2, i3, i4, i5, i6, i7 = [x[..., i] for i in range(8)]
    z1 = (i2 + i6) * _F["f0_541"]
    t2, t3 = z1 - i6 * _F["f1_847"], z1 + i2 * _F["f0_765"]
    t0, t1 = (i0 + i4) << _CB, (i0 - i4) << _CB
    t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    t0, t1, t2, t3 = i7, i5, i3, i1
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * _F["f1_175"]
    t0, t1, t2, t3 = t0 * _F["f0_298"], t1 * _F["f2_053"], t2 * _F["f3_072"], t3 * _F["f1_501"]
    z1, z2, z3, z4 = -z1 * _F["f0_899"], -z2 * _F["f2_562"], -z3 * _F["f1_961"] + z5, -z4 * _F["f0_390"] + z5
    t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
    sh = _CB - _P1 if first else _CB + _P1 + 3
    return np.stack([_ds(t10 + t3, sh), _ds(t11 + t2, sh), _ds(t12 + t1, sh), _ds(t13 + t0, sh), _ds(t13 - t0, sh), _ds(t12 - t1, sh),
                     _ds(t11 - t2, sh), _ds(t10 - t3, sh)], -1)


def _codec_plane(p, table):
    """8-bit plane [H, W] (multiples of 8) through forward DCT, quantisation (division, round half away: jcdctmgr.c), dequantisation,
    inverse DCT and the range limit -- every 8x8 block at once."""
    H, W = p.shape
    b = p.astype(np.int64).reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3) - 128          # [by, bx, row, col]
    c = _fdct8(b, True)                                                     # rows
    c = _fdct8(c.swapaxes(-1, -2), False).swapaxes(-1, -2)                  # columns: coefficients x 8
    qv = table.astype(np.int64) * 8
    lvl = (np.abs(c) + (qv >> 1)) // qv
    c = np.where(c < 0, -lvl, lvl) * table.astype(np.int64)
    w = _idct8(c.swapaxes(-1, -2), True).swapaxes(-1, -2)                   # columns
    r = _idct8(w, False)                                                    # rows
    return np.clip(r + 128, 0, 255).transpose(0, 2, 1, 3).reshape(H, W)


def _fix(x):
    return int(x * 65536 + 0.5)


def jpeg_u8(u8, quality):
    """uint8 RGB [3, H, W] -> the image libjpeg decodes after encoding it at `quality` with its defaults (4:2:0, islow DCT, fancy
    up-sampling) -- what cv2.imencode('.jpg') + cv2.imdecode return (dataops/augmennt/augmennt/extra_functional.py:293-297).  Integer
    arithmetic throughout; equal bit for bit to libjpeg-turbo through PIL (tests/test_degrade.py)."""
    _, H, W = u8.shape
    Hp, Wp = -(-H // 16) * 16, -(-W // 16) * 16
    r, g, b = np.pad(u8.astype(np.int64), ((0, 0), (0, Hp - H), (0, Wp - W)), mode="edge")
    Y = (_fix(0.299) * r + _fix(0.587) * g + _fix(0.114) * b + 32768) >> 16                                  # jccolor.c rgb_ycc_convert
    Cb = (-_fix(0.16874) * r - _fix(0.33126) * g + _fix(0.5) * b + (128 << 16) + 32767) >> 16
    Cr = (_fix(0.5) * r - _fix(0.41869) * g - _fix(0.08131) * b + (128 << 16) + 32767) >> 16
    Hc, Wc = -(-H // 2), -(-W // 2)          # what the decoder's up-sampler sees (downsampled_height / _width)

    def down(p):                              # jcsample.c h2v2_downsample: box average, bias 1, 2, 1, 2, ... along a row;
        s = p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2]
        d = (s + np.where(np.arange(s.shape[1]) % 2 == 0, 1, 2)[None, :]) >> 2
        d[Hc:] = d[Hc - 1]                    # jcprepct.c: below the image the last real DOWN-SAMPLED row is replicated
        return d

    Yq = _codec_plane(Y, _qtable(JQ_LUMA, quality))
    Cbq, Crq = _codec_plane(down(Cb), _qtable(JQ_CHROMA, quality)), _codec_plane(down(Cr), _qtable(JQ_CHROMA, quality))

    def up(p):                                # jdsample.c h2v2_fancy_upsample: (9, 3, 3, 1) / 16, biases 8 (even) / 7 (odd output column)
        p = p[:Hc, :Wc]
        ys, xs = np.mgrid[0:2 * Hc, 0:2 * Wc]
        cy, cx = ys >> 1, xs >> 1
        if Wc <= 2:                           # jdsample.c: the triangle filter needs more than 2 columns, else plain replication
            return p[cy, cx][:H, :W]
        ny = np.clip(cy + np.where(ys & 1, 1, -1), 0, Hc - 1)
        nx = np.clip(cx + np.where(xs & 1, 1, -1), 0, Wc - 1)
        col = lambda c: 3 * p[cy, c] + p[ny, c]
        return ((3 * col(cx) + col(nx) + np.where(xs & 1, 7, 8)) >> 4)[:H, :W]

    yy, cb, cr = Yq[:H, :W], up(Cbq) - 128, up(Crq) - 128
    R = yy + ((_fix(1.402) * cr + 32768) >> 16)                                                              # jdcolor.c ycc_rgb_convert
    B = yy + ((_fix(1.772) * cb + 32768) >> 16)
    G = yy + ((-_fix(0.34414) * cb + 32768 - _fix(0.71414) * cr) >> 16)
    return np.clip(np.stack([R, G, B]), 0, 255).astype(np.uint8)


def jpeg(img, quality):
    """img [3,H,W] RGB in [0,1] -> JPEG round trip of round(255 img) (see jpeg_u8), back in [0,1]."""
    u8 = np.rint(255 * np.clip(img.astype(np.float32), 0, 1)).astype(np.uint8)
    return jpeg_u8(u8, quality).astype(np.float64) / 255.0
