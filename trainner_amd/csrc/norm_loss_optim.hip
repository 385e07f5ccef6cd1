// BatchNorm2d (training mode) + LeakyReLU, the tiny classifier GEMMs, L1 / relativistic-BCE losses,
// global-norm clipping and Adam.  Everything here is HBM- or latency-bound; reductions go through
// double-precision partial sums in a caller-provided workspace and are finalised in a fixed order
// (bit-reproducible run to run).
#include "common.h"

namespace {

constexpr int RED_BLOCKS = 512;  // max partial slabs of any reduction below

__device__ __forceinline__ double block_reduce_sum(double v, double *sh) {
    // 256 threads; result valid in thread 0
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
#pragma unroll
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------------- BatchNorm
// partial[b][c] = {sum f(z), sum g(z)} over the block's pixel range, per channel.
// MODE 0 (fwd):  f = z,            g = z*z
// MODE 1 (bwd):  f = gy*mask(y),   g = gy*mask(y)*(z - mean[c])
// MODE 2 (bwd):  the same with the activation mask RECOMPUTED from z (y is not read): y > 0 <=> ((z - mean) invstd) gamma + beta > 0,
//                the forward's own expression (bn_fwd_apply_kernel; -ffp-contract=off), so the mask is bit-for-bit the one y carries;
//                `y` then points at {invstd[C], gamma[C], beta[C]} pointers packed by the caller: see BnMaskK
struct BnMaskK { const float *invstd, *gamma, *beta; };
template <int MODE>
__global__ void bn_partial_kernel(const float *z, int z_ct, int z_co, const float *gy, int g_ct, int g_co, const float *y,
                                  int y_ct, int y_co, const float *mean, float mslope, int64_t pixels, int C,
                                  int64_t pix_per_block, double *partial, BnMaskK mk = BnMaskK{nullptr, nullptr, nullptr}) {
    extern __shared__ double sh_d[];  // [lanes][C][2]
    {   // blockIdx.y = statistics group (InstanceNorm: one image; BatchNorm launches have one group): `pixels` per group
        const size_t go = (size_t)blockIdx.y * (size_t)pixels;
        z += go * z_ct;
        if (MODE >= 1) {
            gy += go * g_ct;
            if (MODE == 1) y += go * y_ct;
            mean += (size_t)blockIdx.y * C;
        }
        partial += (size_t)blockIdx.y * gridDim.x * C * 2;
    }
    const int G = C / 4;
    const int lanes = 256 / G;
    const int tid = threadIdx.x;
    const int cg = tid % G, pl = tid / G;
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    int64_t p1 = p0 + pix_per_block;
    if (p1 > pixels) p1 = pixels;
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    if (pl < lanes) {
        f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = mu, ga = mu, be = mu;
        if (MODE >= 1) mu = *reinterpret_cast<const f32x4 *>(mean + cg * 4);
        if (MODE == 2) {
            is = *reinterpret_cast<const f32x4 *>(mk.invstd + cg * 4);
            ga = *reinterpret_cast<const f32x4 *>(mk.gamma + cg * 4);
            be = *reinterpret_cast<const f32x4 *>(mk.beta + cg * 4);
        }
        for (int64_t p = p0 + pl; p < p1; p += lanes) {
            const f32x4 zv = *reinterpret_cast<const f32x4 *>(z + p * z_ct + z_co + cg * 4);
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    s0[k] += (double)zv[k];
                    s1[k] += (double)zv[k] * (double)zv[k];
                }
            } else {
                const f32x4 gv = *reinterpret_cast<const f32x4 *>(gy + p * g_ct + g_co + cg * 4);
                f32x4 yv;
                if (MODE == 1) {
                    yv = *reinterpret_cast<const f32x4 *>(y + p * y_ct + y_co + cg * 4);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) yv[k] = ((zv[k] - mu[k]) * is[k]) * ga[k] + be[k];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float g = gv[k] * (yv[k] > 0.f ? 1.f : mslope);
                    s0[k] += (double)g;
                    s1[k] += (double)g * (double)(zv[k] - mu[k]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sh_d[((size_t)pl * C + cg * 4 + k) * 2 + 0] = s0[k];
            sh_d[((size_t)pl * C + cg * 4 + k) * 2 + 1] = s1[k];
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        double a = 0, b = 0;
        for (int l = 0; l < lanes; ++l) {
            a += sh_d[((size_t)l * C + c) * 2 + 0];
            b += sh_d[((size_t)l * C + c) * 2 + 1];
        }
        partial[((size_t)blockIdx.x * C + c) * 2 + 0] = a;
        partial[((size_t)blockIdx.x * C + c) * 2 + 1] = b;
    }
}

// Finalize kernels: 256 threads = 32 channels x 8 lanes; lane l sums partial blocks l, l+8, ... (the
// loads of the 8 lanes are independent, so the ~500 partials cost ~60 dependent steps, not ~500), then
// the lane sums are combined in a fixed order through LDS.
__device__ __forceinline__ void bn_sum_partials(const double *partial, int nblocks, int C, int c, int sl, int el,
                                                double (*sh)[2][33], double &s0, double &s1) {
    double a = 0, b = 0;
    if (c < C)
        for (int q = sl; q < nblocks; q += 8) {
            a += partial[((size_t)q * C + c) * 2 + 0];
            b += partial[((size_t)q * C + c) * 2 + 1];
        }
    sh[sl][0][el] = a;
    sh[sl][1][el] = b;
    __syncthreads();
    s0 = 0;
    s1 = 0;
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        s0 += sh[l][0][el];
        s1 += sh[l][1][el];
    }
}

__global__ void __launch_bounds__(256)
bn_fwd_finalize_kernel(const double *partial, int nblocks, int C, int64_t pixels, float *running_mean, float *running_var,
                       int64_t *num_batches, float momentum, float eps, float *save_mean, float *save_invstd, double *stat64) {
    __shared__ double sh[8][2][33];
    partial += (size_t)blockIdx.y * nblocks * C * 2;       // statistics group (see bn_partial_kernel)
    save_mean += (size_t)blockIdx.y * C;
    save_invstd += (size_t)blockIdx.y * C;
    const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + el;
    double s, ss;
    bn_sum_partials(partial, nblocks, C, c, sl, el, sh, s, ss);
    if (sl == 0 && c < C) {
        const double n = (double)pixels;
        const double mean = s / n;
        double var = ss / n - mean * mean;
        if (var < 0) var = 0;
        save_mean[c] = (float)mean;
        save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        const double unbiased = n > 1 ? var * n / (n - 1) : var;
        if (running_mean) {
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
        }
        if (stat64) {          // the exact values of the running-statistics update (tnr_bn_replay_running re-applies it)
            stat64[c] = mean;
            stat64[C + c] = unbiased;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches) *num_batches += 1;
}

__global__ void bn_fwd_apply_kernel(const float *z, int z_ct, int z_co, float *y, int y_ct, int y_co, int64_t pixels, int C,
                                    const float *gamma, const float *beta, const float *mean, const float *invstd, int act,
                                    float slope) {
    const int G = C / 4;
    const int64_t total = pixels * G;
    z += (size_t)blockIdx.y * (size_t)pixels * z_ct;       // statistics group (see bn_partial_kernel)
    y += (size_t)blockIdx.y * (size_t)pixels * y_ct;
    mean += (size_t)blockIdx.y * C;
    invstd += (size_t)blockIdx.y * C;
    const f32x4 one4 = {1.f, 1.f, 1.f, 1.f}, zero4 = {0.f, 0.f, 0.f, 0.f};
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(e % G);
        const int64_t p = e / G;
        const f32x4 zv = *reinterpret_cast<const f32x4 *>(z + p * z_ct + z_co + cg * 4);
        const f32x4 mu = *reinterpret_cast<const f32x4 *>(mean + cg * 4);
        const f32x4 is = *reinterpret_cast<const f32x4 *>(invstd + cg * 4);
        const f32x4 ga = gamma ? *reinterpret_cast<const f32x4 *>(gamma + cg * 4) : one4;      // (no affine: InstanceNorm)
        const f32x4 be = beta ? *reinterpret_cast<const f32x4 *>(beta + cg * 4) : zero4;
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = tnr_act(((zv[k] - mu[k]) * is[k]) * ga[k] + be[k], act, slope);
        *reinterpret_cast<f32x4 *>(y + p * y_ct + y_co + cg * 4) = o;
    }
}

__global__ void __launch_bounds__(256)
bn_bwd_finalize_kernel(const double *partial, int nblocks, int C, const float *invstd, double *sums, float *dgamma,
                       float *dbeta, float acc_beta) {
    __shared__ double sh[8][2][33];
    partial += (size_t)blockIdx.y * nblocks * C * 2;       // statistics group (see bn_partial_kernel)
    invstd += (size_t)blockIdx.y * C;
    sums += (size_t)blockIdx.y * 2 * C;
    const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + el;
    double s, dot;
    bn_sum_partials(partial, nblocks, C, c, sl, el, sh, s, dot);
    if (sl == 0 && c < C) {
        sums[2 * c + 0] = s;
        sums[2 * c + 1] = dot;
        if (dgamma) {
            const float pg = acc_beta != 0.f ? acc_beta * dgamma[c] : 0.f;
            const float pb = acc_beta != 0.f ? acc_beta * dbeta[c] : 0.f;
            dgamma[c] = pg + (float)(dot * (double)invstd[c]);
            dbeta[c] = pb + (float)s;
        }
    }
}

template <bool FROMZ>      // FROMZ: the activation mask recomputed from z (bn_partial_kernel MODE 2); y is not read, beta is
__global__ void bn_bwd_apply_kernel(const float *gy, int g_ct, int g_co, const float *y, int y_ct, int y_co, const float *z,
                                    int z_ct, int z_co, float *gz, int o_ct, int o_co, int64_t pixels, int C,
                                    const float *gamma, const float *mean, const float *invstd, const double *sums,
                                    float mslope, const float *beta) {
    const int G = C / 4;
    const int64_t total = pixels * G;
    const float inv_n = 1.f / (float)pixels;
    {   // statistics group (see bn_partial_kernel)
        const size_t go = (size_t)blockIdx.y * (size_t)pixels;
        gy += go * g_ct;
        if (!FROMZ) y += go * y_ct;
        z += go * z_ct;
        gz += go * o_ct;
        mean += (size_t)blockIdx.y * C;
        invstd += (size_t)blockIdx.y * C;
        sums += (size_t)blockIdx.y * 2 * C;
    }
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, e0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (stride % G == 0) {
        // a thread keeps its channel group over the whole loop (every power-of-two channel count): the per-channel constants are
        // computed once, with the same expressions as below (bit-identical), and the pixel index advances by a constant --
        // no 64-bit division, no double-precision loads per element (3.8 TB/s before)
        const int cg = (int)(e0 % G);
        float is4[4], gm4[4], kk4[4], mu4[4], ga4[4], be4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = cg * 4 + k;
            be4[k] = FROMZ ? beta[c] : 0.f;
            is4[k] = invstd[c];
            gm4[k] = (float)sums[2 * c + 0] * inv_n;
            kk4[k] = (float)sums[2 * c + 1] * is4[k] * is4[k] * inv_n;
            mu4[k] = mean[c];
            ga4[k] = gamma ? gamma[c] : 1.f;
        }
        const int64_t pstep = stride / G;
        for (int64_t p = e0 / G; p < pixels; p += pstep) {
            const f32x4 gv = *reinterpret_cast<const f32x4 *>(gy + p * g_ct + g_co + cg * 4);
            const f32x4 zv = *reinterpret_cast<const f32x4 *>(z + p * z_ct + z_co + cg * 4);
            f32x4 yv;
            if constexpr (FROMZ) {
#pragma unroll
                for (int k = 0; k < 4; ++k) yv[k] = ((zv[k] - mu4[k]) * is4[k]) * ga4[k] + be4[k];
            } else {
                yv = *reinterpret_cast<const f32x4 *>(y + p * y_ct + y_co + cg * 4);
            }
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float g = gv[k] * (yv[k] > 0.f ? 1.f : mslope);
                o[k] = (g - gm4[k] - (zv[k] - mu4[k]) * kk4[k]) * is4[k] * ga4[k];
            }
            *reinterpret_cast<f32x4 *>(gz + p * o_ct + o_co + cg * 4) = o;
        }
        return;
    }
    for (int64_t e = e0; e < total; e += stride) {
        const int cg = (int)(e % G);
        const int64_t p = e / G;
        const f32x4 gv = *reinterpret_cast<const f32x4 *>(gy + p * g_ct + g_co + cg * 4);
        const f32x4 zv = *reinterpret_cast<const f32x4 *>(z + p * z_ct + z_co + cg * 4);
        f32x4 yv;
        if constexpr (FROMZ) {
#pragma unroll
            for (int k = 0; k < 4; ++k) yv[k] = ((zv[k] - mean[cg * 4 + k]) * invstd[cg * 4 + k]) * (gamma ? gamma[cg * 4 + k] : 1.f) + beta[cg * 4 + k];
        } else {
            yv = *reinterpret_cast<const f32x4 *>(y + p * y_ct + y_co + cg * 4);
        }
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = cg * 4 + k;
            const float is = invstd[c];
            const float gmean = (float)sums[2 * c + 0] * inv_n;
            const float kk = (float)sums[2 * c + 1] * is * is * inv_n;
            const float g = gv[k] * (yv[k] > 0.f ? 1.f : mslope);
            o[k] = (g - gmean - (zv[k] - mean[c]) * kk) * is * (gamma ? gamma[c] : 1.f);
        }
        *reinterpret_cast<f32x4 *>(gz + p * o_ct + o_co + cg * 4) = o;
    }
}

inline void bn_plan(int64_t pixels, int C, int &nblocks, int64_t &ppb, size_t &lds) {
    const int G = C / 4;
    const int lanes = 256 / G;
    int64_t want = tnr_cdiv64(pixels, (int64_t)lanes * 16);
    if (want > RED_BLOCKS) want = RED_BLOCKS;
    if (want < 1) want = 1;
    ppb = tnr_cdiv64(pixels, want);
    nblocks = (int)tnr_cdiv64(pixels, ppb);
    lds = (size_t)lanes * C * 2 * sizeof(double);
}

// ---------------------------------------------------------------------------------- Linear
__global__ void linear_fwd_kernel(const float *x, const float *w, const float *b, float *y, int N, int In, int Out, int act,
                                  float slope) {
    __shared__ double sh[256];
    const int n = blockIdx.x / Out, o = blockIdx.x % Out;
    const float *xr = x + (size_t)n * In, *wr = w + (size_t)o * In;
    float acc = 0.f;
    for (int i = threadIdx.x; i < In; i += 256) acc += xr[i] * wr[i];
    const double tot = block_reduce_sum((double)acc, sh);
    if (threadIdx.x == 0) y[(size_t)n * Out + o] = tnr_act((float)tot + (b ? b[o] : 0.f), act, slope);
}

__global__ void linear_gpre_kernel(const float *gy, const float *yact, float mslope, float *gpre, int64_t n) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) gpre[e] = yact ? gy[e] * (yact[e] > 0.f ? 1.f : mslope) : gy[e];
}

__global__ void linear_bwd_w_kernel(const float *x, const float *gpre, float *dw, float *db, int N, int In, int Out,
                                    float acc_beta) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (int64_t)Out * In) {
        const int o = (int)(e / In), i = (int)(e % In);
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += gpre[(size_t)n * Out + o] * x[(size_t)n * In + i];
        dw[e] = (acc_beta != 0.f ? acc_beta * dw[e] : 0.f) + s;
    }
    if (db && e < Out) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += gpre[(size_t)n * Out + e];
        db[e] = (acc_beta != 0.f ? acc_beta * db[e] : 0.f) + s;
    }
}

__global__ void linear_bwd_x_kernel(const float *w, const float *gpre, float *gx, int N, int In, int Out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (int64_t)N * In) {
        const int n = (int)(e / In), i = (int)(e % In);
        float s = 0.f;
        for (int o = 0; o < Out; ++o) s += gpre[(size_t)n * Out + o] * w[(size_t)o * In + i];
        gx[e] = s;
    }
}

// ---------------------------------------------------------------------------------- losses
__global__ void l1_partial_kernel(const float *a, const float *b, int64_t n, double *partial) {
    __shared__ double sh[256];
    double acc = 0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        acc += (double)fabsf(a[e] - b[e]);
    const double tot = block_reduce_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ void l1_finalize_kernel(const double *partial, int nblocks, int64_t n, float scale, float *loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0;
        for (int b = 0; b < nblocks; ++b) s += partial[b];
        *loss = (float)((double)scale * s / (double)n);
    }
}

__global__ void l1_bwd_kernel(const float *a, const float *b, int64_t n, float scale, const float *gscale, float *ga,
                              int accumulate) {
    const float k = scale * (gscale ? *gscale : 1.f) / (float)n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const float d = a[e] - b[e];
        const float g = d > 0.f ? k : (d < 0.f ? -k : 0.f);
        ga[e] = accumulate ? ga[e] + g : g;
    }
}

__device__ __forceinline__ float softplusf(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void ragan_a_kernel(const float *pf, const float *pr, int n, float *sums) {
    __shared__ double sh[256];
    double sf = 0, sr = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        sf += pf[i];
        sr += pr[i];
    }
    sf = block_reduce_sum(sf, sh);
    sr = block_reduce_sum(sr, sh);
    if (threadIdx.x == 0) {
        sums[0] = (float)sf;
        sums[1] = (float)sr;
        sums[2] = (float)n;
        sums[3] = sums[4] = sums[5] = sums[6] = sums[7] = 0.f;
    }
}

__global__ void ragan_b_kernel(const float *pf, const float *pr, int n, int stage, float *sums) {
    __shared__ double sh[256];
    const float mf = sums[0] / sums[2], mr = sums[1] / sums[2];
    double t1 = 0, t2 = 0, sa = 0, sb = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float dr = pr[i] - mf, df = pf[i] - mr;
        if (stage == 0) {  // generator: BCE(r - mean f, 0) + BCE(f - mean r, 1)
            t1 += softplusf(dr);
            t2 += softplusf(-df);
            sa += sigmoidf(dr);
        } else {  // discriminator: BCE(r - mean f, 1) + BCE(f - mean r, 0)
            t1 += softplusf(-dr);
            t2 += softplusf(df);
            sa += sigmoidf(-dr);
            sb += sigmoidf(df);
        }
    }
    t1 = block_reduce_sum(t1, sh);
    t2 = block_reduce_sum(t2, sh);
    sa = block_reduce_sum(sa, sh);
    sb = block_reduce_sum(sb, sh);
    __syncthreads();
    if (threadIdx.x == 0) {
        sums[3] = (float)t1;
        sums[4] = (float)t2;
        sums[5] = (float)sa;
        sums[6] = (float)sb;
    }
}

// Per-pixel logit maps (UNetDiscriminator, discriminators.py:686-779: n = N x H x W = 4.2 M at batch 16, 512 x 512): the same three
// phases as a two-stage reduction in a FIXED order -- up to 256 blocks write their double partial sums to ws, one block adds them in
// index order -- and an elementwise gradient pass over the whole grid.  (The single-block forms above took 6.8 + 12.6 + 8.1 ms per call
// there: 19 % of the Real-ESRGAN step, profiles/r04k_*.)
constexpr int RAGAN_MULTI_MIN = 4096;       // below: the single-block kernels (bit-for-bit what they always computed)

__global__ void ragan_a_partial_kernel(const float *pf, const float *pr, int n, double *ws) {
    __shared__ double sh[256];
    double sf = 0, sr = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        sf += pf[i];
        sr += pr[i];
    }
    sf = block_reduce_sum(sf, sh);
    sr = block_reduce_sum(sr, sh);
    if (threadIdx.x == 0) {
        ws[2 * blockIdx.x] = sf;
        ws[2 * blockIdx.x + 1] = sr;
    }
}

__global__ void ragan_a_final_kernel(const double *ws, int nb, int n, float *sums) {
    if (threadIdx.x == 0) {
        double sf = 0, sr = 0;
        for (int b = 0; b < nb; ++b) {
            sf += ws[2 * b];
            sr += ws[2 * b + 1];
        }
        sums[0] = (float)sf;
        sums[1] = (float)sr;
        sums[2] = (float)n;
        sums[3] = sums[4] = sums[5] = sums[6] = sums[7] = 0.f;
    }
}

__global__ void ragan_b_partial_kernel(const float *pf, const float *pr, int n, int stage, const float *sums, double *ws) {
    __shared__ double sh[256];
    const float mf = sums[0] / sums[2], mr = sums[1] / sums[2];
    double t1 = 0, t2 = 0, sa = 0, sb = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float dr = pr[i] - mf, df = pf[i] - mr;
        if (stage == 0) {
            t1 += softplusf(dr);
            t2 += softplusf(-df);
            sa += sigmoidf(dr);
        } else {
            t1 += softplusf(-dr);
            t2 += softplusf(df);
            sa += sigmoidf(-dr);
            sb += sigmoidf(df);
        }
    }
    t1 = block_reduce_sum(t1, sh);
    t2 = block_reduce_sum(t2, sh);
    sa = block_reduce_sum(sa, sh);
    sb = block_reduce_sum(sb, sh);
    if (threadIdx.x == 0) {
        double *w = ws + 4 * blockIdx.x;
        w[0] = t1; w[1] = t2; w[2] = sa; w[3] = sb;
    }
}

__global__ void ragan_b_final_kernel(const double *ws, int nb, float *sums) {
    if (threadIdx.x == 0) {
        double t[4] = {0, 0, 0, 0};
        for (int b = 0; b < nb; ++b)
            for (int k = 0; k < 4; ++k) t[k] += ws[4 * b + k];
        sums[3] = (float)t[0];
        sums[4] = (float)t[1];
        sums[5] = (float)t[2];
        sums[6] = (float)t[3];
    }
}

__global__ void ragan_c_grid_kernel(const float *pf, const float *pr, int n, int stage, float weight, const float *sums,
                                    float *loss_out, float *gf, float *gr) {
    const float NN = sums[2];
    const float mf = sums[0] / NN, mr = sums[1] / NN;
    const float l1 = sums[3] / NN, l2 = sums[4] / NN;
    const float sa = sums[5], sb = sums[6];
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        loss_out[0] = weight * ((l1 + l2) * 0.5f);
        loss_out[1] = l1;
        loss_out[2] = l2;
        loss_out[3] = mr;
        loss_out[4] = mf;
    }
    const float k = weight * 0.5f / NN;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float dr = pr[i] - mf, df = pf[i] - mr;
        if (stage == 0) {
            if (gf) gf[i] = k * (-(sa / NN) - sigmoidf(-df));
            if (gr) gr[i] = 0.f;
        } else {
            if (gf) gf[i] = k * (sigmoidf(df) + sa / NN);
            if (gr) gr[i] = k * (-sigmoidf(-dr) - sb / NN);
        }
    }
}

// loss_out: [0] weight * (l1 + l2)/2, [1] l1 (real-side term), [2] l2 (fake-side term), [3] mean r, [4] mean f
__global__ void ragan_c_kernel(const float *pf, const float *pr, int n, int stage, float weight, const float *sums,
                               float *loss_out, float *gf, float *gr) {
    const float NN = sums[2];
    const float mf = sums[0] / NN, mr = sums[1] / NN;
    const float l1 = sums[3] / NN, l2 = sums[4] / NN;
    const float sa = sums[5], sb = sums[6];
    if (threadIdx.x == 0) {
        loss_out[0] = weight * ((l1 + l2) * 0.5f);
        loss_out[1] = l1;
        loss_out[2] = l2;
        loss_out[3] = mr;
        loss_out[4] = mf;
    }
    const float k = weight * 0.5f / NN;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float dr = pr[i] - mf, df = pf[i] - mr;
        if (stage == 0) {
            if (gf) gf[i] = k * (-(sa / NN) - sigmoidf(-df));
            if (gr) gr[i] = 0.f;
        } else {
            if (gf) gf[i] = k * (sigmoidf(df) + sa / NN);
            if (gr) gr[i] = k * (-sigmoidf(-dr) - sb / NN);
        }
    }
}

__global__ void scale_by_kernel(float *dst, const float *src, int64_t n, const float *gscale) {
    const float k = *gscale;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        dst[e] = src[e] * k;
}

// ---------------------------------------------------------------------------------- optimiser
__global__ void sumsq_partial_kernel(const float *g, int64_t n, double *partial) {
    __shared__ double sh[256];
    double acc = 0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        acc += (double)g[e] * (double)g[e];
    const double tot = block_reduce_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ void sumsq_finalize_kernel(const double *partial, int nblocks, double *out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0;
        for (int b = 0; b < nblocks; ++b) s += partial[b];
        *out = s;
    }
}

__global__ void clip_by_norm_kernel(float *g, int64_t n, const double *sumsq, float max_norm) {
    const float total = (float)sqrt(*sumsq);
    float coef = max_norm / (total + 1e-6f);
    if (coef > 1.f) coef = 1.f;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        g[e] *= coef;
}

__global__ void adam_kernel(float *p, const float *g, float *m, float *v, int64_t n, float step_size, float b1, float b2,
                            float bc2_sqrt, float eps, float wd, const unsigned *fault) {
    if (fault != nullptr && *fault != 0u) return;      // a faulted step is never applied (uniform: every thread reads the same word)
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        float gr = g[e];
        const float pv = p[e];
        if (wd != 0.f) gr += wd * pv;
        float mv = m[e], vv = v[e];
        mv = mv + (gr - mv) * (1.f - b1);          // exp_avg.lerp_(grad, 1 - beta1)
        vv = vv * b2 + (1.f - b2) * gr * gr;       // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        m[e] = mv;
        v[e] = vv;
        p[e] = pv - step_size * (mv / denom);      // param.addcdiv_(exp_avg, denom, value=-step_size)
    }
}

inline unsigned grid_for(int64_t n) {
    int64_t b = tnr_cdiv64(n, 256);
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

// ================================================================================== C ABI
extern "C" int64_t tnr_bn_workspace_bytes(int32_t C) { return (int64_t)(RED_BLOCKS + 1) * C * 2 * sizeof(double); }
extern "C" int64_t tnr_reduce_workspace_bytes(void) { return (int64_t)2048 * sizeof(double); }

extern "C" int tnr_bn_train_fwd_stats(tnr_view z, tnr_view y, int64_t pixels, int32_t C, const float *gamma, const float *beta,
                                      float *running_mean, float *running_var, int64_t *num_batches, float momentum, float eps,
                                      float *save_mean, float *save_invstd, double *stat64, int32_t act, float slope, void *ws,
                                      void *stream) {
    TNR_REQUIRE(z.ptr && y.ptr && gamma && beta && save_mean && save_invstd && ws, "bn_fwd: null pointer");
    TNR_REQUIRE((C % 4) == 0 && C <= 1024 && (256 % (C / 4) == 0 || C / 4 > 0), "bn_fwd: unsupported C %d", C);
    TNR_REQUIRE(C / 4 <= 256, "bn_fwd: C too large");
    int nblocks; int64_t ppb; size_t lds;
    bn_plan(pixels, C, nblocks, ppb, lds);
    hipStream_t s = (hipStream_t)stream;
    double *partial = (double *)ws;
    hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(nblocks), dim3(256), lds, s, z.ptr, z.ctot, z.coff, nullptr, 0, 0, nullptr, 0,
                       0, nullptr, 0.f, pixels, C, ppb, partial);
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(tnr_cdiv(C, 32)), dim3(256), 0, s, partial, nblocks, C, pixels,
                       running_mean, running_var, num_batches, momentum, eps, save_mean, save_invstd, stat64);
    hipLaunchKernelGGL(bn_fwd_apply_kernel, dim3(grid_for(pixels * (C / 4))), dim3(256), 0, s, z.ptr, z.ctot, z.coff, y.ptr,
                       y.ctot, y.coff, pixels, C, gamma, beta, save_mean, save_invstd, act, slope);
    return tnr_check_launch("bn_train_fwd");
}

extern "C" int tnr_bn_train_fwd(tnr_view z, tnr_view y, int64_t pixels, int32_t C, const float *gamma, const float *beta,
                                float *running_mean, float *running_var, int64_t *num_batches, float momentum, float eps,
                                float *save_mean, float *save_invstd, int32_t act, float slope, void *ws, void *stream) {
    return tnr_bn_train_fwd_stats(z, y, pixels, C, gamma, beta, running_mean, running_var, num_batches, momentum, eps, save_mean,
                                  save_invstd, nullptr, act, slope, ws, stream);
}

namespace {
__global__ void bn_replay_kernel(float *running_mean, float *running_var, int64_t *num_batches, const double *stat64, int C, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * stat64[c]);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * stat64[C + c]);
    }
    if (c == 0 && num_batches) *num_batches += 1;
}
}  // namespace

// The side effects of ONE MORE training-mode forward over the same batch (running statistics momentum update, batch counter),
// from the statistics tnr_bn_train_fwd_stats recorded: a forward pass whose result is already known (same input, same
// parameters: the discriminator sees the real / fake batch twice per SR step, sr_model.py:170-177,190-193) is not recomputed.
extern "C" int tnr_bn_replay_running(float *running_mean, float *running_var, int64_t *num_batches, const double *stat64, int32_t C,
                                     float momentum, void *stream) {
    TNR_REQUIRE(running_mean && running_var && stat64 && C > 0, "bn_replay_running: bad arguments");
    hipLaunchKernelGGL(bn_replay_kernel, dim3(tnr_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, running_mean, running_var, num_batches,
                       stat64, C, momentum);
    return tnr_check_launch("bn_replay_running");
}

extern "C" int tnr_bn_train_bwd(tnr_view gy, tnr_view y, tnr_view z, tnr_view gz, int64_t pixels, int32_t C,
                                const float *gamma, const float *save_mean, const float *save_invstd, float mslope,
                                float *dgamma, float *dbeta, float acc_beta, void *ws, void *stream) {
    TNR_REQUIRE(gy.ptr && y.ptr && z.ptr && gz.ptr && gamma && save_mean && save_invstd && ws, "bn_bwd: null pointer");
    TNR_REQUIRE((C % 4) == 0 && C / 4 <= 256, "bn_bwd: unsupported C %d", C);
    TNR_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "bn_bwd: dgamma and dbeta go together");
    int nblocks; int64_t ppb; size_t lds;
    bn_plan(pixels, C, nblocks, ppb, lds);
    hipStream_t s = (hipStream_t)stream;
    double *partial = (double *)ws;
    double *sums = partial + (size_t)RED_BLOCKS * C * 2;
    hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(nblocks), dim3(256), lds, s, z.ptr, z.ctot, z.coff, gy.ptr, gy.ctot, gy.coff,
                       y.ptr, y.ctot, y.coff, save_mean, mslope, pixels, C, ppb, partial);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(tnr_cdiv(C, 32)), dim3(256), 0, s, partial, nblocks, C, save_invstd, sums,
                       dgamma, dbeta, acc_beta);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(grid_for(pixels * (C / 4))), dim3(256), 0, s, gy.ptr, gy.ctot, gy.coff, y.ptr,
                       y.ctot, y.coff, z.ptr, z.ctot, z.coff, gz.ptr, gz.ctot, gz.coff, pixels, C, gamma, save_mean,
                       save_invstd, sums, mslope, (const float *)nullptr);
    return tnr_check_launch("bn_train_bwd");
}

extern "C" int tnr_bn_train_bwd_z(tnr_view gy, tnr_view z, tnr_view gz, int64_t pixels, int32_t C, const float *gamma, const float *beta,
                                  const float *save_mean, const float *save_invstd, float mslope, float *dgamma, float *dbeta,
                                  float acc_beta, void *ws, void *stream) {
    TNR_REQUIRE(gy.ptr && z.ptr && gz.ptr && gamma && beta && save_mean && save_invstd && ws, "bn_bwd_z: null pointer");
    TNR_REQUIRE((C % 4) == 0 && C / 4 <= 256, "bn_bwd_z: unsupported C %d", C);
    TNR_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "bn_bwd_z: dgamma and dbeta go together");
    int nblocks; int64_t ppb; size_t lds;
    bn_plan(pixels, C, nblocks, ppb, lds);
    hipStream_t s = (hipStream_t)stream;
    double *partial = (double *)ws;
    double *sums = partial + (size_t)RED_BLOCKS * C * 2;
    hipLaunchKernelGGL(bn_partial_kernel<2>, dim3(nblocks), dim3(256), lds, s, z.ptr, z.ctot, z.coff, gy.ptr, gy.ctot, gy.coff,
                       (const float *)nullptr, 0, 0, save_mean, mslope, pixels, C, ppb, partial, BnMaskK{save_invstd, gamma, beta});
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(tnr_cdiv(C, 32)), dim3(256), 0, s, partial, nblocks, C, save_invstd, sums,
                       dgamma, dbeta, acc_beta);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(grid_for(pixels * (C / 4))), dim3(256), 0, s, gy.ptr, gy.ctot, gy.coff,
                       (const float *)nullptr, 0, 0, z.ptr, z.ctot, z.coff, gz.ptr, gz.ctot, gz.coff, pixels, C, gamma, save_mean,
                       save_invstd, sums, mslope, beta);
    return tnr_check_launch("bn_train_bwd_z");
}

// InstanceNorm2d (no affine, no running statistics: ResNet_arch.py:40-50) forward / backward over a batch in ONE set of
// launches: the BatchNorm kernels with one statistics group per image (gridDim.y = N), optional ReLU / LeakyReLU fused.
extern "C" int64_t tnr_instnorm_workspace_bytes(int32_t N, int32_t C) {
    return (int64_t)N * ((int64_t)RED_BLOCKS * C * 2 + 2 * (int64_t)C) * (int64_t)sizeof(double);
}

extern "C" int tnr_instnorm_fwd(tnr_view z, tnr_view y, int32_t N, int64_t pixels, int32_t C, float eps, float *save_mean,
                                float *save_invstd, int32_t act, float slope, void *ws, void *stream) {
    TNR_REQUIRE(z.ptr && y.ptr && save_mean && save_invstd && ws && N > 0 && N <= 65535 && pixels > 0, "instnorm_fwd: bad arguments");
    TNR_REQUIRE((C % 4) == 0 && C / 4 <= 256, "instnorm_fwd: unsupported C %d", C);
    int nblocks; int64_t ppb; size_t lds;
    bn_plan(pixels, C, nblocks, ppb, lds);
    hipStream_t s = (hipStream_t)stream;
    double *partial = (double *)ws;
    hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(nblocks, N), dim3(256), lds, s, z.ptr, z.ctot, z.coff, nullptr, 0, 0, nullptr, 0,
                       0, nullptr, 0.f, pixels, C, ppb, partial);
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(tnr_cdiv(C, 32), N), dim3(256), 0, s, partial, nblocks, C, pixels,
                       nullptr, nullptr, nullptr, 0.f, eps, save_mean, save_invstd, nullptr);
    int64_t gx = tnr_cdiv64(pixels * (C / 4), 256);
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(bn_fwd_apply_kernel, dim3((unsigned)gx, N), dim3(256), 0, s, z.ptr, z.ctot, z.coff, y.ptr,
                       y.ctot, y.coff, pixels, C, nullptr, nullptr, save_mean, save_invstd, act, slope);
    return tnr_check_launch("instnorm_fwd");
}

extern "C" int tnr_instnorm_bwd(tnr_view gy, tnr_view y, tnr_view z, tnr_view gz, int32_t N, int64_t pixels, int32_t C,
                                const float *save_mean, const float *save_invstd, float mslope, void *ws, void *stream) {
    TNR_REQUIRE(gy.ptr && y.ptr && z.ptr && gz.ptr && save_mean && save_invstd && ws && N > 0 && N <= 65535 && pixels > 0,
                "instnorm_bwd: bad arguments");
    TNR_REQUIRE((C % 4) == 0 && C / 4 <= 256, "instnorm_bwd: unsupported C %d", C);
    int nblocks; int64_t ppb; size_t lds;
    bn_plan(pixels, C, nblocks, ppb, lds);
    hipStream_t s = (hipStream_t)stream;
    double *partial = (double *)ws;
    double *sums = partial + (size_t)N * RED_BLOCKS * C * 2;
    hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(nblocks, N), dim3(256), lds, s, z.ptr, z.ctot, z.coff, gy.ptr, gy.ctot, gy.coff,
                       y.ptr, y.ctot, y.coff, save_mean, mslope, pixels, C, ppb, partial);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(tnr_cdiv(C, 32), N), dim3(256), 0, s, partial, nblocks, C, save_invstd, sums,
                       nullptr, nullptr, 0.f);
    int64_t gx = tnr_cdiv64(pixels * (C / 4), 256);
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3((unsigned)gx, N), dim3(256), 0, s, gy.ptr, gy.ctot, gy.coff, y.ptr,
                       y.ctot, y.coff, z.ptr, z.ctot, z.coff, gz.ptr, gz.ctot, gz.coff, pixels, C, nullptr, save_mean,
                       save_invstd, sums, mslope, (const float *)nullptr);
    return tnr_check_launch("instnorm_bwd");
}

extern "C" int tnr_linear_fwd(const float *x, const float *w, const float *b, float *y, int32_t N, int32_t In, int32_t Out,
                              int32_t act, float slope, void *stream) {
    TNR_REQUIRE(x && w && y && N > 0 && In > 0 && Out > 0, "linear_fwd: bad arguments");
    hipLaunchKernelGGL(linear_fwd_kernel, dim3(N * Out), dim3(256), 0, (hipStream_t)stream, x, w, b, y, N, In, Out, act, slope);
    return tnr_check_launch("linear_fwd");
}

extern "C" int tnr_linear_bwd(const float *x, const float *w, const float *gy, const float *yact, float mslope, float *gx,
                              float *dw, float *db, int32_t N, int32_t In, int32_t Out, float acc_beta, float *gpre_ws,
                              void *stream) {
    TNR_REQUIRE(x && w && gy && gpre_ws, "linear_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(linear_gpre_kernel, dim3(tnr_cdiv(N * Out, 256)), dim3(256), 0, s, gy, yact, mslope, gpre_ws,
                       (int64_t)N * Out);
    if (dw) {
        int64_t tot = (int64_t)Out * In;
        if (tot < Out) tot = Out;
        hipLaunchKernelGGL(linear_bwd_w_kernel, dim3((unsigned)tnr_cdiv64(tot, 256)), dim3(256), 0, s, x, gpre_ws, dw, db, N, In,
                           Out, acc_beta);
    }
    if (gx)
        hipLaunchKernelGGL(linear_bwd_x_kernel, dim3((unsigned)tnr_cdiv64((int64_t)N * In, 256)), dim3(256), 0, s, w, gpre_ws, gx,
                           N, In, Out);
    return tnr_check_launch("linear_bwd");
}

extern "C" int tnr_l1_mean_fwd(const float *a, const float *b, int64_t n, float scale, float *loss, void *ws, void *stream) {
    TNR_REQUIRE(a && b && loss && ws && n > 0, "l1_fwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const unsigned nb = grid_for(n) > 1024 ? 1024 : grid_for(n);
    hipLaunchKernelGGL(l1_partial_kernel, dim3(nb), dim3(256), 0, s, a, b, n, (double *)ws);
    hipLaunchKernelGGL(l1_finalize_kernel, dim3(1), dim3(64), 0, s, (const double *)ws, (int)nb, n, scale, loss);
    return tnr_check_launch("l1_mean_fwd");
}

extern "C" int tnr_l1_mean_bwd(const float *a, const float *b, int64_t n, float scale, const float *gscale, float *ga,
                               int32_t accumulate, void *stream) {
    TNR_REQUIRE(a && b && ga && n > 0, "l1_bwd: bad arguments");
    hipLaunchKernelGGL(l1_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, n, scale, gscale, ga,
                       accumulate);
    return tnr_check_launch("l1_mean_bwd");
}

extern "C" int tnr_ragan_phase_a(const float *pf, const float *pr, int32_t n, float *sums, void *ws, void *stream) {
    TNR_REQUIRE(pf && pr && sums && n > 0, "ragan_a: bad arguments");
    if (n >= RAGAN_MULTI_MIN && ws != nullptr) {
        const int nb = n / 4096 > 256 ? 256 : (n / 4096 < 1 ? 1 : n / 4096);
        hipLaunchKernelGGL(ragan_a_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, pf, pr, n, (double *)ws);
        hipLaunchKernelGGL(ragan_a_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double *)ws, nb, n, sums);
        return tnr_check_launch("ragan_a");
    }
    hipLaunchKernelGGL(ragan_a_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pf, pr, n, sums);
    return tnr_check_launch("ragan_a");
}

extern "C" int tnr_ragan_phase_b(const float *pf, const float *pr, int32_t n, int32_t stage, float *sums, void *ws, void *stream) {
    TNR_REQUIRE(pf && pr && sums && n > 0, "ragan_b: bad arguments");
    if (n >= RAGAN_MULTI_MIN && ws != nullptr) {
        const int nb = n / 4096 > 256 ? 256 : (n / 4096 < 1 ? 1 : n / 4096);
        hipLaunchKernelGGL(ragan_b_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, pf, pr, n, stage, (const float *)sums, (double *)ws);
        hipLaunchKernelGGL(ragan_b_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double *)ws, nb, sums);
        return tnr_check_launch("ragan_b");
    }
    hipLaunchKernelGGL(ragan_b_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pf, pr, n, stage, sums);
    return tnr_check_launch("ragan_b");
}

extern "C" int tnr_ragan_phase_c(const float *pf, const float *pr, int32_t n, int32_t stage, float weight, const float *sums,
                                 float *loss_out, float *gf, float *gr, void *stream) {
    TNR_REQUIRE(pf && pr && sums && loss_out && n > 0, "ragan_c: bad arguments");
    if (n >= RAGAN_MULTI_MIN) {
        hipLaunchKernelGGL(ragan_c_grid_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, pf, pr, n, stage, weight, sums, loss_out, gf, gr);
        return tnr_check_launch("ragan_c");
    }
    hipLaunchKernelGGL(ragan_c_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pf, pr, n, stage, weight, sums, loss_out, gf,
                       gr);
    return tnr_check_launch("ragan_c");
}

extern "C" int tnr_scale_by(float *dst, const float *src, int64_t n, const float *gscale, void *stream) {
    TNR_REQUIRE(dst && src && gscale && n > 0, "scale_by: bad arguments");
    hipLaunchKernelGGL(scale_by_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dst, src, n, gscale);
    return tnr_check_launch("scale_by");
}

extern "C" int tnr_sumsq(const float *g, int64_t n, double *out, void *ws, void *stream) {
    TNR_REQUIRE(g && out && ws && n > 0, "sumsq: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const unsigned nb = grid_for(n);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, s, g, n, (double *)ws);
    hipLaunchKernelGGL(sumsq_finalize_kernel, dim3(1), dim3(64), 0, s, (const double *)ws, (int)nb, out);
    return tnr_check_launch("sumsq");
}

extern "C" int tnr_clip_by_norm(float *g, int64_t n, const double *sumsq, float max_norm, void *stream) {
    TNR_REQUIRE(g && sumsq && n > 0, "clip_by_norm: bad arguments");
    hipLaunchKernelGGL(clip_by_norm_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, g, n, sumsq, max_norm);
    return tnr_check_launch("clip_by_norm");
}

extern "C" int tnr_adam_step(float *p, const float *g, float *m, float *v, int64_t n, float step_size, float b1, float b2,
                             float bc2_sqrt, float eps, float weight_decay, void *stream) {
    TNR_REQUIRE(p && g && m && v && n > 0, "adam: bad arguments");
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, step_size, b1, b2,
                       bc2_sqrt, eps, weight_decay, (const unsigned *)nullptr);
    return tnr_check_launch("adam");
}

extern "C" int tnr_adam_step_guarded(float *p, const float *g, float *m, float *v, int64_t n, float step_size, float b1, float b2,
                                     float bc2_sqrt, float eps, float weight_decay, const uint32_t *fault, void *stream) {
    TNR_REQUIRE(p && g && m && v && n > 0, "adam: bad arguments");
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, step_size, b1, b2,
                       bc2_sqrt, eps, weight_decay, (const unsigned *)fault);
    return tnr_check_launch("adam");
}
