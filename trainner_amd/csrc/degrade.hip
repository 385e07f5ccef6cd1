// Real-ESRGAN style degradations on the device: the reference synthesises the LR image from the HR
// crop per sample on DataLoader worker CPUs with OpenCV (dataops/augmentations.py:1666-1801, presets
// options/presets/resrgan_{blur,resize,noise}.yaml): blur (iso / aniso Gaussian, sinc; 7..21 taps) -> random resize
// (area / linear / cubic) -> Gaussian or Poisson noise (colour or grey) -> JPEG, twice, then the final resize (+ sinc).
// Here the per-pixel work runs as kernels over fp32 NCHW images in [0, 1] (RGB); the random parameters and the small
// blur kernels are drawn on the host (dataops/degradations.py).  Geometry follows OpenCV's published definitions:
//   filter2D    correlation, anchor at the centre, BORDER_REFLECT_101
//   resize      INTER_LINEAR / INTER_CUBIC (A = -0.75) with src = (dst + 0.5) * scale - 0.5 and replicated borders;
//               INTER_AREA = exact box coverage when shrinking on both axes, the "area" variant of linear otherwise
//   JPEG        JFIF YCbCr, 4:2:0, 8x8 DCT, Annex-K tables scaled by libjpeg's quality rule, triangle chroma up-sampling
// Noise uses a counter-based generator (Philox4x32-10 keyed by (seed, sample)): the result depends only on the seed and
// the element index, never on the launch geometry, so tests are reproducible.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ Philox4x32-10
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0, 1)

struct Rng {            // a stream of uniforms for one element: counter = (element, sample, draw block, purpose)
    uint32_t e_lo, e_hi, k0, k1, blk, have;
    uint32_t buf[4];
    __device__ Rng(uint64_t elem, uint32_t sample, uint64_t seed, uint32_t purpose)
        : e_lo((uint32_t)elem), e_hi((uint32_t)(elem >> 32) ^ (purpose << 24)), k0((uint32_t)seed ^ (sample * 0x9E3779B1u)),
          k1((uint32_t)(seed >> 32)), blk(0), have(0) {}
    __device__ float next() {
        if (have == 0) {
            philox4x32_10(e_lo, e_hi, blk++, 0x5EEDu, k0, k1, buf);
            have = 4;
        }
        return u01(buf[--have]);
    }
};

__device__ __forceinline__ float normal01(Rng &g) {   // Box-Muller (one value per call; the pair's second half is dropped)
    const float u1 = g.next(), u2 = g.next();
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// Poisson(lam): sequential search (Knuth) for small means, transformed rejection (PTRS, Hoermann 1993) otherwise
__device__ float poisson(Rng &g, float lam) {
    if (lam <= 0.f) return 0.f;
    if (lam < 10.f) {
        const float L = expf(-lam);
        float p = 1.f;
        int k = 0;
        do {
            ++k;
            p *= g.next();
        } while (p > L && k < 1000);
        return (float)(k - 1);
    }
    const float slam = sqrtf(lam), loglam = logf(lam);
    const float b = 0.931f + 2.53f * slam, a = -0.059f + 0.02483f * b, inv_alpha = 1.1239f + 1.1328f / (b - 3.4f);
    const float vr = 0.9277f - 3.6224f / (b - 2.f);
    for (int it = 0; it < 200; ++it) {
        const float U = g.next() - 0.5f, V = g.next();
        const float us = 0.5f - fabsf(U);
        const float k = floorf((2.f * a / us + b) * U + lam + 0.43f);
        if (us >= 0.07f && V <= vr) return k;
        if (k < 0.f || (us < 0.013f && V > us)) continue;
        if (logf(V) + logf(inv_alpha) - logf(a / (us * us) + b) <= -lam + k * loglam - lgammaf(k + 1.f)) return k;
    }
    return floorf(lam + 0.5f);
}

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// ------------------------------------------------------------------------------------------------ filter2D
constexpr int F_T = 16, F_KMAX = 21, F_R = F_KMAX / 2, F_S = F_T + 2 * F_R;

// grid (tiles_x, tiles_y, N*C); kernels [N][21*21] centred in the 21x21 slot (zeros outside the ks x ks support)
__global__ void __launch_bounds__(256) filter2d_kernel(const float *src, float *dst, const float *kernels, int N, int C, int H, int W) {
    __shared__ float tile[F_S * F_S];
    __shared__ float kw[F_KMAX * F_KMAX];
    const int nc = blockIdx.z, n = nc / C;
    const float *sp = src + (size_t)nc * H * W;
    const int x0 = blockIdx.x * F_T - F_R, y0 = blockIdx.y * F_T - F_R;
    for (int i = threadIdx.x; i < F_KMAX * F_KMAX; i += 256) kw[i] = kernels[(size_t)n * F_KMAX * F_KMAX + i];
    for (int i = threadIdx.x; i < F_S * F_S; i += 256) {
        const int ty = i / F_S, tx = i - ty * F_S;
        tile[i] = sp[(size_t)reflect101(y0 + ty, H) * W + reflect101(x0 + tx, W)];
    }
    __syncthreads();
    const int lx = threadIdx.x % F_T, ly = threadIdx.x / F_T;
    const int ox = blockIdx.x * F_T + lx, oy = blockIdx.y * F_T + ly;
    if (ox >= W || oy >= H) return;
    float acc = 0.f;
    for (int j = 0; j < F_KMAX; ++j)
#pragma unroll
        for (int i = 0; i < F_KMAX; ++i) acc = fmaf(kw[j * F_KMAX + i], tile[(ly + j) * F_S + lx + i], acc);
    dst[(size_t)nc * H * W + (size_t)oy * W + ox] = acc;
}

// ------------------------------------------------------------------------------------------------ resize
__device__ __forceinline__ void cubic_w(float x, float w[4]) {
    const float A = -0.75f;
    w[0] = ((A * (x + 1.f) - 5.f * A) * (x + 1.f) + 8.f * A) * (x + 1.f) - 4.f * A;
    w[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
    w[2] = ((A + 2.f) * (1.f - x) - (A + 3.f)) * (1.f - x) * (1.f - x) + 1.f;
    w[3] = 1.f - w[0] - w[1] - w[2];
}

// one axis of cv::resize's INTER_AREA table for output index d (shrinking: scale = ssize / dsize >= 1): up to `cap` taps
__device__ int area_taps(int d, double scale, int ssize, int *idx, float *wt, int cap) {
    const double fsx1 = d * scale, fsx2 = fsx1 + scale;
    const double cell = fmin(scale, (double)ssize - fsx1);
    int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
    sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
    sx1 = sx1 < sx2 ? sx1 : sx2;
    int k = 0;
    if (sx1 - fsx1 > 1e-3 && k < cap) {
        idx[k] = sx1 - 1;
        wt[k++] = (float)((sx1 - fsx1) / cell);
    }
    for (int sx = sx1; sx < sx2 && k < cap; ++sx) {
        idx[k] = sx;
        wt[k++] = (float)(1.0 / cell);
    }
    if (fsx2 - sx2 > 1e-3 && k < cap) {
        idx[k] = sx2;
        wt[k++] = (float)(fmin(fmin(fsx2 - sx2, 1.0), cell) / cell);
    }
    return k;
}

constexpr int AREA_CAP = 24;    // taps per axis: scale factors up to ~22 (the presets shrink by at most 1 / 0.15 = 6.7 per stage)

// mode 0 area, 1 linear, 2 cubic.  One thread per output element of [NC, Ho, Wo].
__global__ void resize_kernel(const float *src, float *dst, int NC, int H, int W, int Ho, int Wo, int mode) {
    const double sx_ = (double)W / Wo, sy_ = (double)H / Ho;
    const bool area_shrink = mode == 0 && sx_ >= 1.0 && sy_ >= 1.0;
    const int64_t total = (int64_t)NC * Ho * Wo;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int dx = (int)(e % Wo);
        const int64_t q = e / Wo;
        const int dy = (int)(q % Ho);
        const float *sp = src + (q / Ho) * (size_t)H * W;
        float r = 0.f;
        if (area_shrink) {
            int ix[AREA_CAP], iy[AREA_CAP];
            float wx[AREA_CAP], wy[AREA_CAP];
            const int nx = area_taps(dx, sx_, W, ix, wx, AREA_CAP), ny = area_taps(dy, sy_, H, iy, wy, AREA_CAP);
            for (int a = 0; a < ny; ++a) {
                float row = 0.f;
                for (int b = 0; b < nx; ++b) row += wx[b] * sp[(size_t)iy[a] * W + ix[b]];
                r += wy[a] * row;
            }
        } else if (mode == 2) {
            float fx = (float)((dx + 0.5) * sx_ - 0.5), fy = (float)((dy + 0.5) * sy_ - 0.5);
            const int bx = (int)floorf(fx), by = (int)floorf(fy);
            fx -= bx;
            fy -= by;
            float wx[4], wy[4];
            cubic_w(fx, wx);
            cubic_w(fy, wy);
            for (int a = 0; a < 4; ++a) {
                const int yy = min(max(by - 1 + a, 0), H - 1);
                float row = 0.f;
                for (int b = 0; b < 4; ++b) row += wx[b] * sp[(size_t)yy * W + min(max(bx - 1 + b, 0), W - 1)];
                r += wy[a] * row;
            }
        } else {
            float fx, fy;
            int bx, by;
            if (mode == 0) {     // INTER_AREA while enlarging: linear taps with the area coordinate rule
                bx = (int)floor(dx * sx_);
                fx = (float)((dx + 1) - (bx + 1) * ((double)Wo / W));
                fx = fx <= 0.f ? 0.f : fx - floorf(fx);
                by = (int)floor(dy * sy_);
                fy = (float)((dy + 1) - (by + 1) * ((double)Ho / H));
                fy = fy <= 0.f ? 0.f : fy - floorf(fy);
            } else {
                fx = (float)((dx + 0.5) * sx_ - 0.5);
                fy = (float)((dy + 0.5) * sy_ - 0.5);
                bx = (int)floorf(fx);
                by = (int)floorf(fy);
                fx -= bx;
                fy -= by;
            }
            if (bx < 0) { bx = 0; fx = 0.f; }
            if (bx >= W - 1) { bx = W - 1; fx = 0.f; }
            if (by < 0) { by = 0; fy = 0.f; }
            if (by >= H - 1) { by = H - 1; fy = 0.f; }
            const int bx1 = min(bx + 1, W - 1), by1 = min(by + 1, H - 1);
            const float top = (1.f - fx) * sp[(size_t)by * W + bx] + fx * sp[(size_t)by * W + bx1];
            const float bot = (1.f - fx) * sp[(size_t)by1 * W + bx] + fx * sp[(size_t)by1 * W + bx1];
            r = (1.f - fy) * top + fy * bot;
        }
        dst[e] = r;
    }
}

// ------------------------------------------------------------------------------------------------ noise
// Gaussian: x += sigma[n][c] / 255 * N(0,1); grey[n] != 0: one draw per pixel shared by the channels (sigma[n][0]).
__global__ void noise_gaussian_kernel(float *img, int N, int C, int H, int W, const float *sigma, const int32_t *grey, uint64_t seed,
                                      int clip) {
    const int64_t hw = (int64_t)H * W, total = (int64_t)N * C * hw;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = e % hw;
        const int c = (int)((e / hw) % C), n = (int)(e / (hw * C));
        const bool g = grey[n] != 0;
        Rng rng((uint64_t)(g ? p : (int64_t)c * hw + p), (uint32_t)n, seed, 1);
        float v = img[e] + sigma[n * C + (g ? 0 : c)] * (1.0f / 255.0f) * normal01(rng);
        if (clip) v = fminf(fmaxf(v, 0.f), 1.f);
        img[e] = v;
    }
}

// Poisson (extra_functional.py:194-218): noisy = Poisson(x * vals) / vals; x += scale * (noisy - x), the difference reduced
// to its luma (0.299 R + 0.587 G + 0.114 B) when grey[n].  vals[n] = 2^ceil(log2(#distinct 8-bit levels)) from the host.
__global__ void noise_poisson_kernel(float *img, int N, int C, int H, int W, const float *vals, const float *scale,
                                     const int32_t *grey, uint64_t seed, int clip) {
    const int64_t hw = (int64_t)H * W, total = (int64_t)N * hw;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = e % hw;
        const int n = (int)(e / hw);
        float d[4], x[4];
        for (int c = 0; c < C; ++c) {
            Rng rng((uint64_t)((int64_t)c * hw + p), (uint32_t)n, seed, 2);
            x[c] = img[((size_t)n * C + c) * hw + p];
            const float xc = fminf(fmaxf(x[c], 0.f), 1.f);
            d[c] = fminf(fmaxf(poisson(rng, xc * vals[n]) / vals[n], 0.f), 1.f) - xc;
        }
        if (grey[n] && C == 3) {
            const float y = 0.299f * d[0] + 0.587f * d[1] + 0.114f * d[2];
            d[0] = d[1] = d[2] = y;
        }
        for (int c = 0; c < C; ++c) {
            float v = x[c] + scale[n] * d[c];
            if (clip) v = fminf(fmaxf(v, 0.f), 1.f);
            img[((size_t)n * C + c) * hw + p] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ JPEG
__constant__ int JQ_LUMA[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
                                14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                                49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
__constant__ int JQ_CHROMA[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
                                  47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                  99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

// ---- JPEG round trip: an integer-exact emulation of libjpeg (what cv2.imencode + cv2.imdecode run in the reference,
// dataops/augmennt/augmennt/extra_functional.py:293-297): JFIF colour conversion in 16-bit fixed point (jccolor.c / jdcolor.c), 4:2:0 with
// the encoder's 2x2 box average and its alternating rounding bias (jcsample.c h2v2_downsample), the "islow" integer forward / inverse
// DCT (jfdctint.c / jidctint.c, 13-bit constants), quantisation by division with round-half-away (jcdctmgr.c), "fancy" triangle
// chroma up-sampling with the decoder's 8 / 7 biases (jdsample.c h2v2_fancy_upsample).  Pinned bit for bit against libjpeg-turbo
// through PIL (tests/test_degrade.py).
__device__ __forceinline__ int jq_entry(int base, int quality) {      // libjpeg: jpeg_quality_scaling + jpeg_add_quant_table (baseline)
    int q = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
    const int sf = q < 50 ? 5000 / q : 200 - 2 * q;
    int t = (base * sf + 50) / 100;
    return t < 1 ? 1 : (t > 255 ? 255 : t);
}

namespace jdct {
constexpr int CB = 13, P1 = 2;
constexpr int F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299, F1_847 = 15137,
              F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
__device__ __forceinline__ int ds(int x, int n) { return (x + (1 << (n - 1))) >> n; }
// one 8-point forward pass over d[0], d[st], ..., d[7 st] (jfdctint.c; first = the row pass)
__device__ __forceinline__ void fwd8(int *d, int st, bool first) {
    const int d0 = d[0], d1 = d[st], d2 = d[2 * st], d3 = d[3 * st], d4 = d[4 * st], d5 = d[5 * st], d6 = d[6 * st], d7 = d[7 * st];
    int t0 = d0 + d7, t7 = d0 - d7, t1 = d1 + d6, t6 = d1 - d6, t2 = d2 + d5, t5 = d2 - d5, t3 = d3 + d4, t4 = d3 - d4;
    const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    const int sh = first ? CB - P1 : CB + P1;
    d[0] = first ? (t10 + t11) << P1 : ds(t10 + t11, P1);
    d[4 * st] = first ? (t10 - t11) << P1 : ds(t10 - t11, P1);
    int z1 = (t12 + t13) * F0_541;
    d[2 * st] = ds(z1 + t13 * F0_765, sh);
    d[6 * st] = ds(z1 + t12 * (-F1_847), sh);
    z1 = t4 + t7;
    int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
    const int z5 = (z3 + z4) * F1_175;
    t4 *= F0_298; t5 *= F2_053; t6 *= F3_072; t7 *= F1_501;
    z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
    z3 += z5; z4 += z5;
    d[7 * st] = ds(t4 + z1 + z3, sh);
    d[5 * st] = ds(t5 + z2 + z4, sh);
    d[3 * st] = ds(t6 + z2 + z3, sh);
    d[st] = ds(t7 + z1 + z4, sh);
}
// one 8-point inverse pass (jidctint.c; first = the column pass)
__device__ __forceinline__ void inv8(int *d, int st, bool first) {
    const int i0 = d[0], i1 = d[st], i2 = d[2 * st], i3 = d[3 * st], i4 = d[4 * st], i5 = d[5 * st], i6 = d[6 * st], i7 = d[7 * st];
    int z1 = (i2 + i6) * F0_541;
    int t2 = z1 + i6 * (-F1_847), t3 = z1 + i2 * F0_765;
    int t0 = (i0 + i4) << CB, t1 = (i0 - i4) << CB;
    const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    t0 = i7; t1 = i5; t2 = i3; t3 = i1;
    z1 = t0 + t3;
    int z2 = t1 + t2, z3 = t0 + t2, z4 = t1 + t3;
    const int z5 = (z3 + z4) * F1_175;
    t0 *= F0_298; t1 *= F2_053; t2 *= F3_072; t3 *= F1_501;
    z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
    z3 += z5; z4 += z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    const int sh = first ? CB - P1 : CB + P1 + 3;
    d[0] = ds(t10 + t3, sh); d[7 * st] = ds(t10 - t3, sh);
    d[st] = ds(t11 + t2, sh); d[6 * st] = ds(t11 - t2, sh);
    d[2 * st] = ds(t12 + t1, sh); d[5 * st] = ds(t12 - t1, sh);
    d[3 * st] = ds(t13 + t0, sh); d[4 * st] = ds(t13 - t0, sh);
}
}  // namespace jdct

// one workgroup (64 threads) per 8x8 block of one plane (Y at full size, Cb / Cr at half size; 8-bit sample values held in floats):
// forward DCT -> quantise -> dequantise -> inverse DCT -> range limit, all in libjpeg's integer arithmetic
__global__ void __launch_bounds__(64) jpeg_quant_kernel(float *plane, int N, int Hp, int Wp, const int32_t *quality, int chroma) {
    __shared__ int blk[64];
    const int t = threadIdx.x, bx = blockIdx.x, by = blockIdx.y, n = blockIdx.z;
    float *p = plane + ((size_t)n * Hp + by * 8 + (t >> 3)) * Wp + bx * 8 + (t & 7);
    blk[t] = (int)*p - 128;
    __syncthreads();
    if (t < 8) jdct::fwd8(blk + 8 * t, 1, true);           // rows
    __syncthreads();
    if (t < 8) jdct::fwd8(blk + t, 8, false);              // columns: coefficients scaled by 8
    __syncthreads();
    {
        const int q = jq_entry(chroma ? JQ_CHROMA[t] : JQ_LUMA[t], quality[n]), qv = q << 3;
        const int c = blk[t], a = (c < 0 ? -c : c) + (qv >> 1);
        const int lvl = a / qv;
        blk[t] = (c < 0 ? -lvl : lvl) * q;                 // quantised, then dequantised by the decoder
    }
    __syncthreads();
    if (t < 8) jdct::inv8(blk + t, 8, true);               // columns
    __syncthreads();
    if (t < 8) jdct::inv8(blk + 8 * t, 1, false);          // rows
    __syncthreads();
    const int v = blk[t] + 128;
    *p = (float)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

__device__ __forceinline__ int jfix(double x) { return (int)(x * 65536.0 + 0.5); }

// RGB [0,1] -> 8-bit Y (full size) and Cb / Cr (half size).  Right edge: the last column is replicated at full resolution; bottom edge:
// the last REAL row of every component is replicated (for the chroma planes that is the last down-sampled row: jcprepct.c)
__global__ void jpeg_to_ycc_kernel(const float *img, int N, int H, int W, float *Y, float *Cb, float *Cr, int Hp, int Wp) {
    const int Hc = Hp / 2, Wc = Wp / 2, Hcv = (H + 1) / 2;
    const int64_t total = (int64_t)N * Hc * Wc;
    const int fy_r = jfix(0.299), fy_g = jfix(0.587), fy_b = jfix(0.114), fb_r = jfix(0.16874), fb_g = jfix(0.33126), f_half = jfix(0.5),
              fr_g = jfix(0.41869), fr_b = jfix(0.08131);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int cx = (int)(e % Wc), cy = (int)((e / Wc) % Hc), n = (int)(e / ((int64_t)Wc * Hc));
        const int cys = cy < Hcv ? cy : Hcv - 1;          // chroma rows below the image repeat the last real one
        int sb = 0, sr = 0;
        for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx) {
                const int x = min(2 * cx + dx, W - 1);
                const size_t pl = (size_t)H * W;
                auto px = [&](int y, int c) {          // the codec sees 8-bit samples: round(255 x)
                    return (int)rintf(255.f * fminf(fmaxf(img[((size_t)n * 3 + c) * pl + (size_t)y * W + x], 0.f), 1.f));
                };
                {
                    const int y = min(2 * cy + dy, H - 1);
                    Y[((size_t)n * Hp + 2 * cy + dy) * Wp + 2 * cx + dx] =
                        (float)((fy_r * px(y, 0) + fy_g * px(y, 1) + fy_b * px(y, 2) + 32768) >> 16);
                }
                const int y = min(2 * cys + dy, H - 1);
                const int r = px(y, 0), g = px(y, 1), b = px(y, 2);
                sb += (-fb_r * r - fb_g * g + f_half * b + (128 << 16) + 32767) >> 16;
                sr += (f_half * r - fr_g * g - fr_b * b + (128 << 16) + 32767) >> 16;
            }
        const int bias = (cx & 1) ? 2 : 1;
        Cb[((size_t)n * Hc + cy) * Wc + cx] = (float)((sb + bias) >> 2);
        Cr[((size_t)n * Hc + cy) * Wc + cx] = (float)((sr + bias) >> 2);
    }
}
// decoded planes -> RGB [0,1]; chroma up-sampled with the triangle ("fancy") filter of libjpeg's h2v2 decoder over the
// ceil(H / 2) x ceil(W / 2) samples the decoder sees
__global__ void jpeg_from_ycc_kernel(float *img, int N, int H, int W, const float *Y, const float *Cb, const float *Cr, int Hp, int Wp) {
    const int Hc = Hp / 2, Wc = Wp / 2, Hcv = (H + 1) / 2, Wcv = (W + 1) / 2;
    const int64_t total = (int64_t)N * H * W;
    const int fr = jfix(1.402), fb = jfix(1.772), fg_b = jfix(0.34414), fg_r = jfix(0.71414);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(e % W), y = (int)((e / W) % H), n = (int)(e / ((int64_t)W * H));
        const int cx = x >> 1, cy = y >> 1;
        const int nx = min(max(cx + ((x & 1) ? 1 : -1), 0), Wcv - 1), ny = min(max(cy + ((y & 1) ? 1 : -1), 0), Hcv - 1);
        const float *pb = Cb + (size_t)n * Hc * Wc, *pr = Cr + (size_t)n * Hc * Wc;
        auto tri = [&](const float *p) {
            if (Wcv <= 2) return (int)p[(size_t)cy * Wc + cx];      // jdsample.c: <= 2 chroma columns -> plain replication
            const int th = 3 * (int)p[(size_t)cy * Wc + cx] + (int)p[(size_t)ny * Wc + cx];       // this column: 3 x near row + far row
            const int ot = 3 * (int)p[(size_t)cy * Wc + nx] + (int)p[(size_t)ny * Wc + nx];       // the neighbouring column
            return (3 * th + ot + ((x & 1) ? 7 : 8)) >> 4;
        };
        const int yy = (int)Y[((size_t)n * Hp + y) * Wp + x];
        const int cb = tri(pb) - 128, cr = tri(pr) - 128;
        const int r = yy + ((fr * cr + 32768) >> 16), b = yy + ((fb * cb + 32768) >> 16), g = yy + ((-fg_b * cb + 32768 - fg_r * cr) >> 16);
        const size_t pl = (size_t)H * W, o = (size_t)y * W + x;
        img[(size_t)n * 3 * pl + o] = (float)min(max(r, 0), 255) * (1.f / 255.f);
        img[((size_t)n * 3 + 1) * pl + o] = (float)min(max(g, 0), 255) * (1.f / 255.f);
        img[((size_t)n * 3 + 2) * pl + o] = (float)min(max(b, 0), 255) * (1.f / 255.f);
    }
}

inline unsigned grid1d(int64_t n) {
    int64_t b = tnr_cdiv64(n, 256);
    return (unsigned)(b > 65535 ? 65535 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int tnr_filter2d(const float *src, float *dst, const float *kernels21, int32_t N, int32_t C, int32_t H, int32_t W, void *stream) {
    TNR_REQUIRE(src && dst && kernels21 && src != dst && N >= 1 && C >= 1 && H >= 1 && W >= 1 && (int64_t)N * C <= 65535,
                "filter2d: bad arguments");
    hipLaunchKernelGGL(filter2d_kernel, dim3(tnr_cdiv(W, F_T), tnr_cdiv(H, F_T), N * C), dim3(256), 0, (hipStream_t)stream, src, dst,
                       kernels21, N, C, H, W);
    return tnr_check_launch("filter2d");
}

extern "C" int tnr_resize(const float *src, float *dst, int32_t NC, int32_t H, int32_t W, int32_t Ho, int32_t Wo, int32_t mode, void *stream) {
    TNR_REQUIRE(src && dst && src != dst && NC >= 1 && H >= 1 && W >= 1 && Ho >= 1 && Wo >= 1 && mode >= 0 && mode <= 2, "resize: bad arguments");
    TNR_REQUIRE(mode != 0 || ((double)W / Wo < AREA_CAP - 2 && (double)H / Ho < AREA_CAP - 2), "resize: area shrink factor too large");
    hipLaunchKernelGGL(resize_kernel, dim3(grid1d((int64_t)NC * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, src, dst, NC, H, W, Ho, Wo, mode);
    return tnr_check_launch("resize");
}

extern "C" int tnr_noise_gaussian(float *img, int32_t N, int32_t C, int32_t H, int32_t W, const float *sigma255, const int32_t *grey,
                                  uint64_t seed, int32_t clip, void *stream) {
    TNR_REQUIRE(img && sigma255 && grey && N >= 1 && C >= 1 && C <= 4, "noise_gaussian: bad arguments");
    hipLaunchKernelGGL(noise_gaussian_kernel, dim3(grid1d((int64_t)N * C * H * W)), dim3(256), 0, (hipStream_t)stream, img, N, C, H, W,
                       sigma255, grey, seed, clip);
    return tnr_check_launch("noise_gaussian");
}

extern "C" int tnr_noise_poisson(float *img, int32_t N, int32_t C, int32_t H, int32_t W, const float *vals, const float *scale,
                                 const int32_t *grey, uint64_t seed, int32_t clip, void *stream) {
    TNR_REQUIRE(img && vals && scale && grey && N >= 1 && C >= 1 && C <= 4, "noise_poisson: bad arguments");
    hipLaunchKernelGGL(noise_poisson_kernel, dim3(grid1d((int64_t)N * H * W)), dim3(256), 0, (hipStream_t)stream, img, N, C, H, W, vals,
                       scale, grey, seed, clip);
    return tnr_check_launch("noise_poisson");
}

extern "C" int64_t tnr_jpeg_workspace_bytes(int32_t N, int32_t H, int32_t W) {
    const int64_t Hp = tnr_round_up(H, 16), Wp = tnr_round_up(W, 16);
    return (int64_t)N * (Hp * Wp + 2 * (Hp / 2) * (Wp / 2)) * (int64_t)sizeof(float);
}

extern "C" int tnr_jpeg_sim(float *img, int32_t N, int32_t H, int32_t W, const int32_t *quality, float *ws, int64_t ws_bytes, void *stream) {
    TNR_REQUIRE(img && quality && ws && N >= 1 && N <= 65535 && H >= 1 && W >= 1, "jpeg_sim: bad arguments");
    TNR_REQUIRE(ws_bytes >= tnr_jpeg_workspace_bytes(N, H, W), "jpeg_sim: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int Hp = tnr_round_up(H, 16), Wp = tnr_round_up(W, 16), Hc = Hp / 2, Wc = Wp / 2;
    float *Y = ws, *Cb = Y + (size_t)N * Hp * Wp, *Cr = Cb + (size_t)N * Hc * Wc;
    hipLaunchKernelGGL(jpeg_to_ycc_kernel, dim3(grid1d((int64_t)N * Hc * Wc)), dim3(256), 0, s, img, N, H, W, Y, Cb, Cr, Hp, Wp);
    hipLaunchKernelGGL(jpeg_quant_kernel, dim3(Wp / 8, Hp / 8, N), dim3(64), 0, s, Y, N, Hp, Wp, quality, 0);
    hipLaunchKernelGGL(jpeg_quant_kernel, dim3(Wc / 8, Hc / 8, N), dim3(64), 0, s, Cb, N, Hc, Wc, quality, 1);
    hipLaunchKernelGGL(jpeg_quant_kernel, dim3(Wc / 8, Hc / 8, N), dim3(64), 0, s, Cr, N, Hc, Wc, quality, 1);
    hipLaunchKernelGGL(jpeg_from_ycc_kernel, dim3(grid1d((int64_t)N * H * W)), dim3(256), 0, s, img, N, H, W, Y, Cb, Cr, Hp, Wp);
    return tnr_check_launch("jpeg_sim");
}
