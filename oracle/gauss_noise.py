"""CPU restatement of the ESRGAN+ noise multiplier field (TEST INFRASTRUCTURE ONLY).

The reference (codes/models/modules/architectures/block.py:587-600) draws `normal_()` from torch's global
generator; the engine draws a counter-based field inside the convolution epilogues
(trainner_amd/csrc/gauss_noise.h).  This file restates THAT function in numpy -- the integer hash bit for
bit, the Box-Muller transform in float32 with libm instead of the hardware's v_log / v_sqrt / v_sin /
v_cos (a few ulp apart) -- for the tests that check the device field against its documentation and
for tests/emul_backend.py.  Nothing under trainner_amd/ imports it.
"""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def _mul(a, c):
    return (a.astype(np.uint64) * np.uint64(c) & M32).astype(np.uint32)


def hash_pair(q, k0, k1):
    """The two 32-bit words of element quad q (uint32 array) under key (k0, k1): tnr_gauss_mult4's integer part."""
    a = q.astype(np.uint32) ^ np.uint32(k0)
    a = _mul(a, 0x9E3779B1)
    a ^= a >> np.uint32(15)
    a = ((a.astype(np.uint64) + np.uint64(k1)) & M32).astype(np.uint32)
    a = _mul(a, 0x85EBCA77)
    a ^= a >> np.uint32(13)
    a = _mul(a, 0xC2B2AE3D)
    a ^= a >> np.uint32(16)
    b = _mul(a ^ np.uint32(k1), 0x27D4EB2F)
    b ^= b >> np.uint32(15)
    b = _mul(b, 0x165667B1)
    b ^= b >> np.uint32(16)
    return a, b


def _pair(h):
    u1 = (h >> np.uint32(16)).astype(np.float32) * np.float32(2.0 ** -16) + np.float32(2.0 ** -17)
    u2 = (h & np.uint32(0xFFFF)).astype(np.float32) * np.float32(2.0 ** -16)
    r = np.sqrt(np.float32(-1.3862943611198906) * np.log2(u1), dtype=np.float32)
    ang = (np.float32(2.0 * np.pi) * u2).astype(np.float32)
    return r * np.cos(ang, dtype=np.float32), r * np.sin(ang, dtype=np.float32)


def normals(pixels, C, k0, k1, pix0=0):
    """n[p, c] of the field: [pixels, C] float32 (C % 4 == 0)."""
    c4 = C // 4
    q = ((np.arange(pixels, dtype=np.uint64)[:, None] + np.uint64(pix0)) * np.uint64(c4) + np.arange(c4, dtype=np.uint64)[None, :]) & M32
    a, b = hash_pair(q.astype(np.uint32), k0, k1)
    n0, n1 = _pair(a)
    n2, n3 = _pair(b)
    return np.stack([n0, n1, n2, n3], axis=-1).reshape(pixels, C).astype(np.float32)


def multiplier(pixels, C, sigma, k0, k1, pix0=0):
    """m = 1 + sigma * n (block.py:597-599: x + n * (sigma * x) = x * m)."""
    return (np.float32(1.0) + np.float32(sigma) * normals(pixels, C, k0, k1, pix0)).astype(np.float32)


def noise_key(seed, call, block):
    """(key0, key1) of (seed, training forward `call`, dense block): trainner_amd.ops.noise_key restated (splitmix64 chain)."""
    M = 0xFFFFFFFFFFFFFFFF

    def mix(z):
        z = (z + 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)

    k = mix(mix(mix(seed & M) ^ (call & M)) ^ (block & M))
    return k & 0xFFFFFFFF, (k >> 32) & 0xFFFFFFFF


def normals_nchw(N, C, H, W, seed, call, block, sample0=0):
    """torch [N, C, H, W] float32: the N(0, 1) field the engine draws for dense block `block` in its `call`-th training forward."""
    import torch
    k0, k1 = noise_key(seed, call, block)
    n = normals(N * H * W, C, k0, k1, sample0 * H * W)
    return torch.from_numpy(n).view(N, H, W, C).permute(0, 3, 1, 2).contiguous()
