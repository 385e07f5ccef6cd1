// ESRGAN+ GaussianNoise (codes/models/modules/architectures/block.py:587-600, used by ResidualDenseBlock_5C.forward,
// RRDBNet_arch.py:160-163; on by default: options/defaults.py:59):   y = x + n * (sigma * x),  n ~ N(0, 1) per element,
// gradient through both terms  =>  y = x * m,  dy/dx = m,  m = 1 + sigma * n.
//
// The multiplier is never stored: it is a pure function of (key, element) and is regenerated where it is needed -- in the epilogue
// of the dense block's last convolution (forward) and in the epilogue that writes the gradient of a noised tensor (backward).
// Counter-based: element quad q = (pixel + pix0) * (C / 4) + channel / 4 (pixel = n * H * W + y * W + x of the LOCAL batch, pix0 =
// first pixel of this rank's shard in the global batch, so that N ranks draw what one process would draw on the concatenated
// batch), key = 64 bits the host derives from (seed, training forward, block).  One call yields the four multipliers of a float4.
// Cost matters (it sits in the epilogue of the dominant kernel, 2.3 G normals per step): two 32-bit hashes per quad, each split
// into a 16-bit radius and a 16-bit angle (Box-Muller on the hardware log / sqrt / sin / cos: |n| <= 4.85, 65 536 angles) -- a
// regulariser's noise, not a Monte-Carlo stream.  The hardware transcendentals are deterministic: forward and backward see the
// same bits as long as both go through THIS function.
#pragma once
#include "common.h"

__device__ __forceinline__ f32x4 tnr_gauss_mult4(const unsigned q, const unsigned k0, const unsigned k1, const float sigma) {
    // keyed at two depths: with the key only xor-ed into the counter, two blocks' fields would be permutations of each other
    unsigned a = q ^ k0;
    a *= 0x9E3779B1u;
    a ^= a >> 15;
    a += k1;
    a *= 0x85EBCA77u;
    a ^= a >> 13;
    a *= 0xC2B2AE3Du;
    a ^= a >> 16;
    unsigned b = (a ^ k1) * 0x27D4EB2Fu;
    b ^= b >> 15;
    b *= 0x165667B1u;
    b ^= b >> 16;
    f32x4 m;
    auto pair = [&](const unsigned h, float &n0, float &n1) {
        const float u1 = __builtin_fmaf((float)(h >> 16), 0x1p-16f, 0x1p-17f);       // (0, 1): radius
        const float u2 = (float)(h & 0xFFFFu) * 0x1p-16f;                            // [0, 1): angle in revolutions
        const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));       // sqrt(-2 ln u1), v_log_f32 = log2
        n0 = r * __builtin_amdgcn_cosf(u2);                                          // v_cos_f32 / v_sin_f32 take revolutions
        n1 = r * __builtin_amdgcn_sinf(u2);
    };
    float n0, n1, n2, n3;
    pair(a, n0, n1);
    pair(b, n2, n3);
    m[0] = __builtin_fmaf(sigma, n0, 1.f);
    m[1] = __builtin_fmaf(sigma, n1, 1.f);
    m[2] = __builtin_fmaf(sigma, n2, 1.f);
    m[3] = __builtin_fmaf(sigma, n3, 1.f);
    return m;
}
