// Weight gradient of a 3x3 s1 p1 layer that has <= 4 channels on one side (the RGB-image layers: D conv0 and
// G's first conv, 3 -> 64; G's last conv, 64 -> 3) on the vector ALUs.  The matrix-core kernel pads the 3 channels
// to a 32-wide block and runs 1.3 ms per launch at 16 x 512 x 512; the real work is 27 FMAs per element of the
// 64-channel operand, so the launch is bounded by reading that operand once (1.07 GB).
//   big   : the >= 16-channel operand (g for a 3 -> C layer, x for a C -> 3 layer), read as float4 of 4 channels
//   small : the 4-channel NHWC4 operand (3 valid), three rows at a time in LDS
//   dW[big channel][small channel][tap] = sum_pixels big[p] * small[p + tap]      (flip = 0: big = g, small = x)
//                                       = sum_pixels big[p] * small[p - tap]      (flip = 1: big = x, small = g)
// One workgroup walks a contiguous range of image rows; thread = (pixel lane, 4 big channels) keeps 4 x 27
// accumulators; pixel lanes are combined by shuffles + LDS in a fixed order, workgroups by a reduce launch over
// [workgroup] partial slabs (deterministic).
#include "common.h"

namespace {

constexpr int WT_MAXW = 512;     // pixels of a row staged per pass
constexpr int WT_GRID = 1024;

struct WThinK {
    const float *big; int b_ct, b_co, Cb;
    const float *small;              // NHWC4
    int N, H, W, flip;
    float *ws;                       // [grid][Cb][28]: 27 weight partials + 1 big-side bias partial
    float *ws_small;                 // [grid][4]: small-side bias partials
    int rows_total, rows_per_wg;
};

template <int TPP>   // threads per pixel = Cb / 4
__global__ void __launch_bounds__(256) wgrad_thin_kernel(const WThinK a) {
    constexpr int PL = 256 / TPP;                              // pixel lanes
    __shared__ __attribute__((aligned(16))) float s_small[3 * (WT_MAXW + 2) * 4];
    __shared__ float s_red[4 * TPP * 112];
    const int tid = threadIdx.x;
    const int c4 = tid % TPP, pl = tid / TPP;
    float acc[4][27];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 27; ++j) acc[k][j] = 0.f;
    f32x4 accb = {0.f, 0.f, 0.f, 0.f};                         // sum of the big operand (bias of a 3 -> C layer)
    f32x4 accs = {0.f, 0.f, 0.f, 0.f};                         // sum of the small operand (bias of a C -> 3 layer)
    const int r_begin = blockIdx.x * a.rows_per_wg;
    int r_end = r_begin + a.rows_per_wg;
    if (r_end > a.rows_total) r_end = a.rows_total;
    for (int row = r_begin; row < r_end; ++row) {
        const int n = row / a.H, y = row - n * a.H;
        for (int x0 = 0; x0 < a.W; x0 += WT_MAXW) {
            const int wc = (a.W - x0 < WT_MAXW) ? a.W - x0 : WT_MAXW;
            __syncthreads();
            for (int i = tid; i < 3 * (wc + 2); i += 256) {     // small rows y-1 .. y+1, columns x0-1 .. x0+wc
                const int rr = i / (wc + 2), cc = i - rr * (wc + 2);
                const int Y = y + rr - 1, X = x0 + cc - 1;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (Y >= 0 && Y < a.H && X >= 0 && X < a.W) v = *reinterpret_cast<const f32x4 *>(a.small + (((size_t)n * a.H + Y) * a.W + X) * 4);
                *reinterpret_cast<f32x4 *>(s_small + (rr * (WT_MAXW + 2) + cc) * 4) = v;
            }
            __syncthreads();
            for (int x = pl; x < wc; x += PL) {
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(a.big + (((size_t)n * a.H + y) * a.W + x0 + x) * a.b_ct + a.b_co + c4 * 4);
                accb += bv;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int ty = t / 3, tx = t - ty * 3;
                    const int rr = a.flip ? 2 - ty : ty, cc = a.flip ? 2 - tx : tx;
                    const f32x4 sv = *reinterpret_cast<const f32x4 *>(s_small + (rr * (WT_MAXW + 2) + x + cc) * 4);
                    if (t == 4 && c4 == 0) accs += sv;          // centre tap: every pixel exactly once
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int ci = 0; ci < 3; ++ci) acc[k][t * 3 + ci] = __builtin_fmaf(bv[k], sv[ci], acc[k][t * 3 + ci]);
                }
            }
        }
    }
    // ---- combine the pixel lanes: inside a wave by xor-shuffles (lanes with equal c4 are TPP apart), then the
    // 4 waves through LDS in wave order
#pragma unroll
    for (int off = TPP; off < 64; off <<= 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int j = 0; j < 27; ++j) acc[k][j] += __shfl_xor(acc[k][j], off);
            accb[k] += __shfl_xor(accb[k], off);
            accs[k] += __shfl_xor(accs[k], off);
        }
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    if (lane < TPP) {
        float *dst = s_red + (wave * TPP + lane) * 112;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int j = 0; j < 27; ++j) dst[k * 27 + j] = acc[k][j];
            dst[108 + k] = accb[k];
        }
    }
    __shared__ float s_accs[4][4];
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s_accs[wave][k] = accs[k];
    }
    __syncthreads();
    // threads 0 .. TPP*112-1: one output each
    for (int i = tid; i < TPP * 112; i += 256) {
        const float v = ((s_red[i] + s_red[TPP * 112 + i]) + s_red[2 * TPP * 112 + i]) + s_red[3 * TPP * 112 + i];
        const int c4o = i / 112, j = i - c4o * 112;
        // slab layout [Cb][28]: channel = 4*c4o + k
        if (j < 108) a.ws[((size_t)blockIdx.x * a.Cb + c4o * 4 + j / 27) * 28 + j % 27] = v;
        else a.ws[((size_t)blockIdx.x * a.Cb + c4o * 4 + (j - 108)) * 28 + 27] = v;
    }
    if (tid < 4) a.ws_small[(size_t)blockIdx.x * 4 + tid] = ((s_accs[0][tid] + s_accs[1][tid]) + s_accs[2][tid]) + s_accs[3][tid];
}

struct WThinRedK {
    const float *ws; const float *ws_small; int grid, Cb, Cs, flip;
    float *dw; float *db; float alpha, beta;
};

// one thread per (big channel, j): j < 27 weight element, j == 27 bias of the big side; plus Cs threads for the
// small-side bias.  Workgroup partials are added in index order, 8 interleaved lanes would not pay here (<= 1024 terms).
__global__ void __launch_bounds__(256) wgrad_thin_reduce_kernel(const WThinRedK a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nmain = a.Cb * 28;
    if (i < nmain) {
        const int cb = i / 28, j = i - cb * 28;
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
        int g = 0;
        for (; g + 3 < a.grid; g += 4) {
            p0 += a.ws[((size_t)g * a.Cb + cb) * 28 + j];
            p1 += a.ws[((size_t)(g + 1) * a.Cb + cb) * 28 + j];
            p2 += a.ws[((size_t)(g + 2) * a.Cb + cb) * 28 + j];
            p3 += a.ws[((size_t)(g + 3) * a.Cb + cb) * 28 + j];
        }
        for (; g < a.grid; ++g) p0 += a.ws[((size_t)g * a.Cb + cb) * 28 + j];
        const float sum = (p0 + p1) + (p2 + p3);
        if (j < 27) {
            const int t = j / 3, cs = j - t * 3;
            if (cs < a.Cs) {
                // flip = 0: dW[cb][cs][t] of a Cs -> Cb layer;  flip = 1: dW[cs][cb][t] of a Cb -> Cs layer
                const size_t o = a.flip ? ((size_t)cs * a.Cb + cb) * 9 + t : ((size_t)cb * a.Cs + cs) * 9 + t;
                a.dw[o] = (a.beta != 0.f ? a.beta * a.dw[o] : 0.f) + a.alpha * sum;
            }
        } else if (!a.flip && a.db != nullptr) {
            a.db[cb] = (a.beta != 0.f ? a.beta * a.db[cb] : 0.f) + a.alpha * sum;
        }
    } else if (i < nmain + a.Cs && a.flip && a.db != nullptr) {
        const int cs = i - nmain;
        float sum = 0.f;
        for (int g = 0; g < a.grid; ++g) sum += a.ws_small[(size_t)g * 4 + cs];
        a.db[cs] = (a.beta != 0.f ? a.beta * a.db[cs] : 0.f) + a.alpha * sum;
    }
}

int thin_grid(int rows) { return rows < WT_GRID ? rows : WT_GRID; }

// ---- 7x7 form (ResnetGenerator's image-side layers, ResNet_arch.py:52-55, :86-88): the 49 taps in one launch.
//   acc[cb][cs][t] = sum over the big grid q of big(q)[cb] * small[q + t + off][cs]
// thread = (pixel lane, ONE big channel) with 49 x 3 accumulators (a tap's 3 image channels: one packed FMA + one scalar); the 7 rows of the
// 4-channel operand a big row meets sit in LDS and are read as broadcasts (the lanes of a pixel share the address); the wide operand
// is read once, through the reflection map when the layer's input is ReflectionPad2d(3)-padded -- no padded copy of it exists.
// Pixel lanes are combined by shuffles (inside a wave) and through LDS in wave order, workgroups by a reduce launch (deterministic).
constexpr int W7_MAXW = 320;     // big-grid columns staged per pass
constexpr int W7_GRID = 512;     // (<= 2 workgroups per CU at ~200 VGPRs: one round)
constexpr int W7_SW = W7_MAXW + 6 + 8;       // (+ the zero columns a row's last pixel group may read)
#ifndef W7_GROUP
#define W7_GROUP 4
#endif
constexpr int W7_G = W7_GROUP;          // adjacent pixels per trip of a pixel lane
typedef float f32x2_t __attribute__((ext_vector_type(2)));

struct WThin7K {
    const float *big; int b_ct, b_co, Cb;
    int N, H, W, rpad;               // the wide buffer and the reflection width of the big grid ((H + 2 rpad) x (W + 2 rpad))
    const float *small;              // NHWC4, Hs x Ws
    int Hs, Ws, off;
    float *ws;                       // [grid][Cb][148]: 147 weight partials + 1 big-side bias partial
    int rows_total, rows_per_wg;
};

template <int TPP>   // threads per pixel = Cb
__global__ void __launch_bounds__(256, 2) wgrad_thin7_kernel(const WThin7K a) {
    constexpr int PL = 256 / TPP;
    __shared__ __attribute__((aligned(16))) float s_small[7 * W7_SW * 4];      // 36.5 KB; reused for the wave reduction (TPP x 98 floats)
    static_assert(7 * W7_SW * 4 >= 64 * 98, "wave-reduction buffer");
    const int tid = threadIdx.x;
    const int c = tid % TPP, pl = tid / TPP;
    f32x2_t acc01[49];
    float acc2[49];               // (the third image channel: a scalar FMA costs what a half-used packed one does, and 49 registers less)
#pragma unroll
    for (int t = 0; t < 49; ++t) {
        acc01[t] = (f32x2_t){0.f, 0.f};
        acc2[t] = 0.f;
    }
    float accb = 0.f;
    const int Hg = a.H + 2 * a.rpad, Wg = a.W + 2 * a.rpad;
    const int r_begin = blockIdx.x * a.rows_per_wg;
    int r_end = r_begin + a.rows_per_wg;
    if (r_end > a.rows_total) r_end = a.rows_total;
    for (int row = r_begin; row < r_end; ++row) {
        const int n = row / Hg, yq = row - n * Hg;
        int ysrc = yq - a.rpad;
        ysrc = ysrc < 0 ? -ysrc : (ysrc >= a.H ? 2 * a.H - 2 - ysrc : ysrc);
        const float *brow = a.big + ((size_t)n * a.H + ysrc) * a.W * a.b_ct + a.b_co + c;
        for (int x0 = 0; x0 < Wg; x0 += W7_MAXW) {
            const int wc = (Wg - x0 < W7_MAXW) ? Wg - x0 : W7_MAXW;
            __syncthreads();
            const int sc = wc + 6 + W7_G - 1;                   // (+ W7_G - 1 zero columns: the last group of a row may hang over it)
            for (int i = tid; i < 7 * sc; i += 256) {           // small rows yq + off .. + 6, columns x0 + off .. x0 + off + wc + 5
                const int rr = i / sc, cc = i - rr * sc;
                const int Y = yq + rr + a.off, X = x0 + cc + a.off;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (cc < wc + 6 && Y >= 0 && Y < a.Hs && X >= 0 && X < a.Ws) v = *reinterpret_cast<const f32x4 *>(a.small + (((size_t)n * a.Hs + Y) * a.Ws + X) * 4);
                *reinterpret_cast<f32x4 *>(s_small + (rr * W7_SW + cc) * 4) = v;
            }
            __syncthreads();
            // a pixel lane takes W7_G ADJACENT pixels per trip: their 7 x 7 windows of the thin operand overlap, so a trip reads
            // 7 x (W7_G + 6) float4 from LDS instead of W7_G x 49 (every lane of a pixel reads the same address, yet such a read costs
            // the LDS pipe as much as any other: at one pixel per trip the pipe, not the vector ALUs, set the pace).  The next trip's
            // values of the wide operand are in flight meanwhile (a wave holds ~200 registers: no other wave hides that latency).
            // Pixels beyond the row count as zeros (the staged columns beyond the row are zeros too).
            auto big_at = [&](int x) -> float {
                if (x >= wc) return 0.f;
                int xsrc = x0 + x - a.rpad;
                xsrc = xsrc < 0 ? -xsrc : (xsrc >= a.W ? 2 * a.W - 2 - xsrc : xsrc);
                return brow[(size_t)xsrc * a.b_ct];
            };
            float cur[W7_G], nxt[W7_G];
#pragma unroll
            for (int j = 0; j < W7_G; ++j) cur[j] = big_at(pl * W7_G + j);
            for (int xg = pl * W7_G; xg < wc; xg += PL * W7_G) {
#pragma unroll
                for (int j = 0; j < W7_G; ++j) nxt[j] = big_at(xg + PL * W7_G + j);
#pragma unroll
                for (int j = 0; j < W7_G; ++j) accb += cur[j];
#pragma unroll
                for (int tr = 0; tr < 7; ++tr) {
#pragma unroll
                    for (int col = 0; col < W7_G + 6; ++col) {
                        const f32x4 sv = *reinterpret_cast<const f32x4 *>(s_small + (tr * W7_SW + xg + col) * 4);
#pragma unroll
                        for (int j = 0; j < W7_G; ++j) {
                            const int tc = col - j;
                            if (tc >= 0 && tc < 7) {
                                const int t = tr * 7 + tc;
                                acc01[t] = __builtin_elementwise_fma((f32x2_t){cur[j], cur[j]}, (f32x2_t){sv[0], sv[1]}, acc01[t]);
                                acc2[t] = __builtin_fmaf(cur[j], sv[2], acc2[t]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < W7_G; ++j) cur[j] = nxt[j];
            }
        }
    }
    // ---- pixel lanes of a wave (lanes with equal c are TPP apart)
#pragma unroll
    for (int off = TPP; off < 64; off <<= 1) {
#pragma unroll
        for (int t = 0; t < 49; ++t) {
            acc01[t][0] += __shfl_xor(acc01[t][0], off);
            acc01[t][1] += __shfl_xor(acc01[t][1], off);
            acc2[t] += __shfl_xor(acc2[t], off);
        }
        accb += __shfl_xor(accb, off);
    }
    // ---- the 4 waves, in wave order, through LDS (two halves: channels 0,1 | channel 2 and the bias sum)
    const int wave = tid >> 6, lane = tid & 63;
    float *red = s_small;
#pragma unroll 1
    for (int w = 1; w < 4; ++w) {
        __syncthreads();
        if (wave == w && lane < TPP) {
#pragma unroll
            for (int t = 0; t < 49; ++t) {
                red[(2 * t) * TPP + lane] = acc01[t][0];
                red[(2 * t + 1) * TPP + lane] = acc01[t][1];
            }
        }
        __syncthreads();
        if (wave == 0 && lane < TPP) {
#pragma unroll
            for (int t = 0; t < 49; ++t) {
                acc01[t][0] += red[(2 * t) * TPP + lane];
                acc01[t][1] += red[(2 * t + 1) * TPP + lane];
            }
        }
        __syncthreads();
        if (wave == w && lane < TPP) {
#pragma unroll
            for (int t = 0; t < 49; ++t) red[t * TPP + lane] = acc2[t];
            red[49 * TPP + lane] = accb;
        }
        __syncthreads();
        if (wave == 0 && lane < TPP) {
#pragma unroll
            for (int t = 0; t < 49; ++t) acc2[t] += red[t * TPP + lane];
            accb += red[49 * TPP + lane];
        }
    }
    if (wave == 0 && lane < TPP) {
        float *dst = a.ws + ((size_t)blockIdx.x * a.Cb + lane) * 148;
#pragma unroll
        for (int t = 0; t < 49; ++t) {
            dst[3 * t] = acc01[t][0];
            dst[3 * t + 1] = acc01[t][1];
            dst[3 * t + 2] = acc2[t];
        }
        dst[147] = accb;
    }
}

struct WThin7RedK {
    const float *ws; int grid, Cb, Cs, flip;
    float *dw; float *db; float alpha, beta;
};

// one thread per (big channel, j): j < 147 weight element (tap j / 3, small channel j % 3), j == 147 the big-side bias
__global__ void __launch_bounds__(256) wgrad_thin7_reduce_kernel(const WThin7RedK a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.Cb * 148) return;
    const int cb = i / 148, j = i - cb * 148;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    int g = 0;
    for (; g + 3 < a.grid; g += 4) {
        p0 += a.ws[((size_t)g * a.Cb + cb) * 148 + j];
        p1 += a.ws[((size_t)(g + 1) * a.Cb + cb) * 148 + j];
        p2 += a.ws[((size_t)(g + 2) * a.Cb + cb) * 148 + j];
        p3 += a.ws[((size_t)(g + 3) * a.Cb + cb) * 148 + j];
    }
    for (; g < a.grid; ++g) p0 += a.ws[((size_t)g * a.Cb + cb) * 148 + j];
    const float sum = (p0 + p1) + (p2 + p3);
    if (j < 147) {
        const int t = j / 3, cs = j - t * 3;
        if (cs < a.Cs) {
            const size_t o = a.flip ? ((size_t)cs * a.Cb + cb) * 49 + (48 - t) : ((size_t)cb * a.Cs + cs) * 49 + t;
            a.dw[o] = (a.beta != 0.f ? a.beta * a.dw[o] : 0.f) + a.alpha * sum;
        }
    } else if (!a.flip && a.db != nullptr) {
        a.db[cb] = (a.beta != 0.f ? a.beta * a.db[cb] : 0.f) + a.alpha * sum;
    }
}

int thin7_grid(int rows) { return rows < W7_GRID ? rows : W7_GRID; }

}  // namespace

extern "C" int64_t tnr_wgrad_thin7_workspace_bytes(int32_t N, int32_t Hgrid, int32_t Cbig) {
    return (int64_t)thin7_grid(N * Hgrid) * Cbig * 148 * (int64_t)sizeof(float);
}

extern "C" int tnr_wgrad_thin7(tnr_view big, int32_t N, int32_t H, int32_t W, int32_t rpad, tnr_view small, int32_t Hs, int32_t Ws, int32_t off,
                               int32_t Cbig, int32_t Csmall, int32_t flip, float *dw, float *db, float alpha, float beta, float *ws,
                               int64_t ws_bytes, void *stream) {
    TNR_REQUIRE(big.ptr && small.ptr && dw && ws && N > 0 && H > 0 && W > 0 && Hs > 0 && Ws > 0, "wgrad_thin7: bad arguments");
    TNR_REQUIRE(Cbig == 16 || Cbig == 32 || Cbig == 64, "wgrad_thin7: the wide side must have 16, 32 or 64 channels (got %d)", Cbig);
    TNR_REQUIRE(Csmall >= 1 && Csmall <= 3 && small.ctot == 4 && small.coff == 0, "wgrad_thin7: the thin side is an NHWC4 image with <= 3 channels");
    TNR_REQUIRE(rpad >= 0 && rpad <= 3 && H > rpad && W > rpad && off >= -6 && off <= 0, "wgrad_thin7: rpad in [0, 3] (< H, W), off in [-6, 0]");
    TNR_REQUIRE(!(flip && db != nullptr), "wgrad_thin7: flip = 1 has no bias output (sum the 4-channel gradient instead)");
    const int Hg = H + 2 * rpad;
    TNR_REQUIRE(tnr_wgrad_thin7_workspace_bytes(N, Hg, Cbig) <= ws_bytes, "wgrad_thin7: workspace too small");
    TNR_REQUIRE((int64_t)N * H * W * big.ctot < (1LL << 40) && (int64_t)N * Hg < (1LL << 31), "wgrad_thin7: buffer too large");
    WThin7K k;
    k.big = big.ptr; k.b_ct = big.ctot; k.b_co = big.coff; k.Cb = Cbig;
    k.N = N; k.H = H; k.W = W; k.rpad = rpad;
    k.small = small.ptr; k.Hs = Hs; k.Ws = Ws; k.off = off;
    const int grid = thin7_grid(N * Hg);
    k.ws = ws;
    k.rows_total = N * Hg; k.rows_per_wg = tnr_cdiv(N * Hg, grid);
    const int used = tnr_cdiv(N * Hg, k.rows_per_wg);
    hipStream_t s = (hipStream_t)stream;
    switch (Cbig) {
        case 16: hipLaunchKernelGGL(wgrad_thin7_kernel<16>, dim3(used), dim3(256), 0, s, k); break;
        case 32: hipLaunchKernelGGL(wgrad_thin7_kernel<32>, dim3(used), dim3(256), 0, s, k); break;
        default: hipLaunchKernelGGL(wgrad_thin7_kernel<64>, dim3(used), dim3(256), 0, s, k); break;
    }
    int rc = tnr_check_launch("wgrad_thin7");
    if (rc != TNR_OK) return rc;
    WThin7RedK r;
    r.ws = ws; r.grid = used; r.Cb = Cbig; r.Cs = Csmall; r.flip = flip ? 1 : 0;
    r.dw = dw; r.db = db; r.alpha = alpha; r.beta = beta;
    hipLaunchKernelGGL(wgrad_thin7_reduce_kernel, dim3(tnr_cdiv(Cbig * 148, 256)), dim3(256), 0, s, r);
    return tnr_check_launch("wgrad_thin7_reduce");
}

extern "C" int64_t tnr_wgrad_thin_workspace_bytes(int32_t N, int32_t H, int32_t Cbig) {
    const int64_t g = thin_grid(N * H);
    return (g * Cbig * 28 + g * 4) * (int64_t)sizeof(float);
}

extern "C" int tnr_wgrad_thin(tnr_view big, tnr_view small, int32_t N, int32_t H, int32_t W, int32_t Cbig, int32_t Csmall, int32_t flip,
                              float *dw, float *db, float alpha, float beta, float *ws, int64_t ws_bytes, void *stream) {
    TNR_REQUIRE(big.ptr && small.ptr && dw && ws && N > 0 && H > 0 && W > 0, "wgrad_thin: bad arguments");
    TNR_REQUIRE(Cbig == 16 || Cbig == 32 || Cbig == 64, "wgrad_thin: the wide side must have 16, 32 or 64 channels (got %d)", Cbig);
    TNR_REQUIRE(Csmall >= 1 && Csmall <= 3 && small.ctot == 4 && small.coff == 0, "wgrad_thin: the thin side is an NHWC4 image with <= 3 channels");
    TNR_REQUIRE((big.ctot % 4) == 0 && (big.coff % 4) == 0, "wgrad_thin: wide view must be 4-channel aligned");
    TNR_REQUIRE(tnr_wgrad_thin_workspace_bytes(N, H, Cbig) <= ws_bytes, "wgrad_thin: workspace too small");
    TNR_REQUIRE((int64_t)N * H * W * big.ctot < (1LL << 40), "wgrad_thin: buffer too large");
    WThinK k;
    k.big = big.ptr; k.b_ct = big.ctot; k.b_co = big.coff; k.Cb = Cbig;
    k.small = small.ptr; k.N = N; k.H = H; k.W = W; k.flip = flip ? 1 : 0;
    const int grid = thin_grid(N * H);
    k.ws = ws; k.ws_small = ws + (size_t)grid * Cbig * 28;
    k.rows_total = N * H; k.rows_per_wg = tnr_cdiv(N * H, grid);
    const int used = tnr_cdiv(N * H, k.rows_per_wg);        // workgroups that actually own rows
    hipStream_t s = (hipStream_t)stream;
    switch (Cbig) {
        case 16: hipLaunchKernelGGL(wgrad_thin_kernel<4>, dim3(used), dim3(256), 0, s, k); break;
        case 32: hipLaunchKernelGGL(wgrad_thin_kernel<8>, dim3(used), dim3(256), 0, s, k); break;
        default: hipLaunchKernelGGL(wgrad_thin_kernel<16>, dim3(used), dim3(256), 0, s, k); break;
    }
    int rc = tnr_check_launch("wgrad_thin");
    if (rc != TNR_OK) return rc;
    WThinRedK r;
    r.ws = ws; r.ws_small = k.ws_small; r.grid = used; r.Cb = Cbig; r.Cs = Csmall; r.flip = k.flip;
    r.dw = dw; r.db = db; r.alpha = alpha; r.beta = beta;
    hipLaunchKernelGGL(wgrad_thin_reduce_kernel, dim3(tnr_cdiv(Cbig * 28 + 4, 256)), dim3(256), 0, s, r);
    return tnr_check_launch("wgrad_thin_reduce");
}
